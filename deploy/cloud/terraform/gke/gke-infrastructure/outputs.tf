data "google_client_config" "me" {}

resource "local_sensitive_file" "kubeconfig" {
  filename        = var.kubeconfig_path
  file_permission = "0600"
  content = yamlencode({
    apiVersion      = "v1"
    kind            = "Config"
    current-context = var.cluster_name
    clusters = [{
      name = var.cluster_name
      cluster = {
        server                     = "https://${google_container_cluster.gke.endpoint}"
        certificate-authority-data = google_container_cluster.gke.master_auth[0].cluster_ca_certificate
      }
    }]
    contexts = [{ name = var.cluster_name, context = { cluster = var.cluster_name, user = var.cluster_name } }]
    users = [{
      name = var.cluster_name
      user = { exec = { apiVersion = "client.authentication.k8s.io/v1beta1", command = "gke-gcloud-auth-plugin", provideClusterInfo = true } }
    }]
  })
}

output "cluster_name" { value = google_container_cluster.gke.name }
output "kubeconfig" { value = local_sensitive_file.kubeconfig.filename }
output "gpu_nodes" { value = "${var.gpu_node_count} x ${var.gpu_machine_type}" }
