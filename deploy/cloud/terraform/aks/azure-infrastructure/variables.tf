variable "subscription_id" { type = string }
variable "resource_group" {
  type    = string
  default = "vpsa-rg"
}
variable "location" {
  type    = string
  default = "westus3"
}
variable "cluster_name" {
  type    = string
  default = "vpsa-aks"
}
variable "kubernetes_version" {
  type    = string
  default = null # the region's default
}
variable "system_vm_size" {
  type    = string
  default = "Standard_D8s_v5"
}
variable "gpu_vm_size" {
  description = "AMD Instinct VM size; ND MI300X v5 carries eight GPUs"
  type        = string
  default     = "Standard_ND96isr_MI300X_v5"
}
variable "gpu_node_count" {
  type    = number
  default = 1
}
variable "gpu_os_disk_gb" {
  type    = number
  default = 1024
}
variable "kubeconfig_path" {
  description = "where the cluster's kubeconfig is written (read by ../production-stack)"
  type        = string
  default     = "../kubeconfig"
}
variable "node_admin_user" {
  description = "admin user on the nodes; its SSH key is made in ssh.tf"
  type        = string
  default     = "azureuser"
}
