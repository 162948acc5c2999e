resource "google_container_node_pool" "system" {
  name       = "system"
  cluster    = google_container_cluster.gke.name
  location   = var.zone
  node_count = 1
  node_config {
    machine_type = var.system_machine_type
    oauth_scopes = ["https://www.googleapis.com/auth/cloud-platform"]
  }
}

resource "google_container_node_pool" "gpu" {
  name       = "gpu"
  cluster    = google_container_cluster.gke.name
  location   = var.zone
  node_count = var.gpu_node_count
  node_config {
    machine_type = var.gpu_machine_type
    disk_size_gb = var.gpu_disk_gb
    disk_type    = "pd-ssd"
    labels       = { accelerator = "amd-instinct" }
    taint {
      key    = "sku"
      value  = "gpu"
      effect = "NO_SCHEDULE"
    }
    oauth_scopes = ["https://www.googleapis.com/auth/cloud-platform"]
  }
  management {
    auto_repair  = true
    auto_upgrade = false # a node upgrade restarts every engine on it: do it when you choose
  }
}
