resource "google_container_cluster" "gke" {
  name     = var.cluster_name
  location = var.zone

  # node pools are managed as their own resources
  remove_default_node_pool = true
  initial_node_count       = 1
  deletion_protection      = false

  release_channel { channel = "REGULAR" }
  workload_identity_config { workload_pool = "${var.project}.svc.id.goog" }
  ip_allocation_policy {}

  depends_on = [google_project_service.container, google_project_service.compute]
}
