# SSH access to the nodes (debugging a GPU node: `dmesg | grep amdgpu`, `rocm-smi`, kubelet logs).
# The key pair is made by Terraform and lives in its state: the public half goes into the
# cluster's linux_profile, the private half is an output to be written to a file
# (`terraform output -raw ssh_private_key > aks_node_key && chmod 600 aks_node_key`), and a
# public-key resource records it in the resource group so the portal shows which key is installed.
# Role of /root/reference/tutorials/terraform/aks/azure-infrastructure/ssh.tf:1-31 (which asks the
# Azure API to generate the pair; here the `tls` provider does, so no azapi provider is needed).
resource "tls_private_key" "node_ssh" {
  algorithm = "RSA" # AKS accepts RSA keys only
  rsa_bits  = 4096
}

resource "azurerm_ssh_public_key" "node_ssh" {
  name                = "${var.cluster_name}-node-key"
  resource_group_name = azurerm_resource_group.rg.name
  location            = azurerm_resource_group.rg.location
  public_key          = tls_private_key.node_ssh.public_key_openssh
}

output "key_data" {
  value = tls_private_key.node_ssh.public_key_openssh
}

output "ssh_private_key" {
  description = "private key for the node admin user (var.node_admin_user); save it to a file to SSH into nodes"
  value       = tls_private_key.node_ssh.private_key_openssh
  sensitive   = true
}
