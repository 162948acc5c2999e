resource "local_sensitive_file" "kubeconfig" {
  content         = azurerm_kubernetes_cluster.aks.kube_config_raw
  filename        = var.kubeconfig_path
  file_permission = "0600"
}
