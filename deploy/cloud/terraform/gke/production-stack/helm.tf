# AMD device plugin + node labeller, then the serving chart from this repository.
locals {
  plugin_base = "https://raw.githubusercontent.com/ROCm/k8s-device-plugin/master"
}

data "http" "device_plugin" { url = "${local.plugin_base}/k8s-ds-amdgpu-dp.yaml" }
data "http" "labeller" { url = "${local.plugin_base}/k8s-ds-amdgpu-labeller.yaml" }

resource "kubernetes_manifest" "device_plugin" {
  manifest = yamldecode(data.http.device_plugin.response_body)
}

resource "kubernetes_manifest" "labeller" {
  # the labeller file holds several documents (RBAC + DaemonSet): one manifest each
  for_each = { for i, d in split("\n---\n", data.http.labeller.response_body) : i => d if trimspace(d) != "" }
  manifest = yamldecode(each.value)
}

resource "helm_release" "stack" {
  name       = var.release
  namespace  = var.namespace
  chart      = "${path.module}/../../../../helm"
  values     = [file(var.values_file)]
  timeout    = 1800
  wait       = true
  depends_on = [kubernetes_manifest.device_plugin]

  # the GPU pool of ../gke-infrastructure is tainted
  set {
    name  = "servingEngineSpec.tolerations[0].key"
    value = "sku"
  }
  set {
    name  = "servingEngineSpec.tolerations[0].operator"
    value = "Equal"
  }
  set {
    name  = "servingEngineSpec.tolerations[0].value"
    value = "gpu"
  }
  set {
    name  = "servingEngineSpec.tolerations[0].effect"
    value = "NoSchedule"
  }
  dynamic "set_sensitive" {
    for_each = var.hf_token == "" ? [] : [var.hf_token]
    content {
      name  = "servingEngineSpec.modelSpec[0].hf_token"
      value = set_sensitive.value
    }
  }
}

output "router_service" {
  value = "${var.release}-router-service.${var.namespace}.svc:80"
}
