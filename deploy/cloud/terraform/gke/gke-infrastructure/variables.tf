variable "project" { type = string }
variable "region" {
  type    = string
  default = "us-central1"
}
variable "zone" {
  type    = string
  default = "us-central1-a"
}
variable "cluster_name" {
  type    = string
  default = "vpsa-gke"
}
variable "system_machine_type" {
  type    = string
  default = "e2-standard-8"
}
variable "gpu_machine_type" {
  description = "machine type of the GPU pool (an AMD Instinct type available to the project)"
  type        = string
}
variable "gpu_node_count" {
  type    = number
  default = 1
}
variable "gpu_disk_gb" {
  type    = number
  default = 500
}
variable "kubeconfig_path" {
  type    = string
  default = "../kubeconfig"
}
