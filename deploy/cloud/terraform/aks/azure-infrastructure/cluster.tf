resource "azurerm_resource_group" "rg" {
  name     = var.resource_group
  location = var.location
}

resource "azurerm_kubernetes_cluster" "aks" {
  name                = var.cluster_name
  location            = azurerm_resource_group.rg.location
  resource_group_name = azurerm_resource_group.rg.name
  dns_prefix          = var.cluster_name
  kubernetes_version  = var.kubernetes_version

  default_node_pool {
    name       = "system"
    node_count = 1
    vm_size    = var.system_vm_size
  }
  identity { type = "SystemAssigned" }
  network_profile { network_plugin = "azure" }
  linux_profile {
    admin_username = var.node_admin_user
    ssh_key { key_data = tls_private_key.node_ssh.public_key_openssh }
  }
}

# GPU pool: tainted so that only workloads that tolerate sku=gpu land there.  AKS installs no
# AMD driver; the node image (or the AMD GPU operator) provides amdgpu, the device plugin of
# ../production-stack makes amd.com/gpu allocatable.
resource "azurerm_kubernetes_cluster_node_pool" "gpu" {
  name                  = "gpupool"
  kubernetes_cluster_id = azurerm_kubernetes_cluster.aks.id
  vm_size               = var.gpu_vm_size
  node_count            = var.gpu_node_count
  os_disk_size_gb       = var.gpu_os_disk_gb
  node_taints           = ["sku=gpu:NoSchedule"]
  node_labels           = { accelerator = "amd-instinct" }
  gpu_driver            = "None"
}
