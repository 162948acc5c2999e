output "cluster_name" { value = azurerm_kubernetes_cluster.aks.name }
output "resource_group" { value = azurerm_resource_group.rg.name }
output "kubeconfig" { value = local_sensitive_file.kubeconfig.filename }
output "gpu_nodes" { value = "${var.gpu_node_count} x ${var.gpu_vm_size}" }
