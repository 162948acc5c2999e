# Installs the AMD GPU device plugin and the serving chart into an existing cluster.
#   terraform init && terraform apply -var kubeconfig=~/.kube/config -var values_file=../../examples/values-01-minimal.yaml
terraform {
  required_providers {
    helm       = { source = "hashicorp/helm", version = ">= 2.12" }
    kubernetes = { source = "hashicorp/kubernetes", version = ">= 2.25" }
    http       = { source = "hashicorp/http", version = ">= 3.4" }
  }
}

variable "kubeconfig" { type = string }
variable "values_file" { type = string }
variable "release" {
  type    = string
  default = "vllm"
}
variable "namespace" {
  type    = string
  default = "default"
}

provider "kubernetes" { config_path = var.kubeconfig }
provider "helm" {
  kubernetes { config_path = var.kubeconfig }
}

data "http" "amd_device_plugin" {
  url = "https://raw.githubusercontent.com/ROCm/k8s-device-plugin/master/k8s-ds-amdgpu-dp.yaml"
}

resource "kubernetes_manifest" "amd_device_plugin" {
  manifest = yamldecode(data.http.amd_device_plugin.response_body)
}

resource "helm_release" "stack" {
  name       = var.release
  namespace  = var.namespace
  chart      = "${path.module}/../../helm"
  values     = [file(var.values_file)]
  timeout    = 1800
  depends_on = [kubernetes_manifest.amd_device_plugin]
}

output "router_service" {
  value = "${var.release}-router-service.${var.namespace}.svc:80"
}
