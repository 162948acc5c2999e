terraform {
  required_version = ">= 1.5"
  required_providers {
    helm       = { source = "hashicorp/helm", version = "~> 2.14" }
    kubernetes = { source = "hashicorp/kubernetes", version = "~> 2.31" }
    http       = { source = "hashicorp/http", version = "~> 3.4" }
  }
}

provider "kubernetes" { config_path = var.kubeconfig }
provider "helm" {
  kubernetes { config_path = var.kubeconfig }
}
