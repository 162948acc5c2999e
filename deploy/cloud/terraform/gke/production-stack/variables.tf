variable "kubeconfig" {
  type    = string
  default = "../kubeconfig"
}
variable "values_file" {
  type    = string
  default = "../production_stack_specification.yaml"
}
variable "release" {
  type    = string
  default = "vllm"
}
variable "namespace" {
  type    = string
  default = "default"
}
variable "hf_token" {
  type      = string
  default   = ""
  sensitive = true
}
