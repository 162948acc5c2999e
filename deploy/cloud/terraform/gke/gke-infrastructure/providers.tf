terraform {
  required_version = ">= 1.5"
  required_providers {
    google = { source = "hashicorp/google", version = "~> 5.40" }
    local  = { source = "hashicorp/local", version = "~> 2.5" }
  }
}

provider "google" {
  project = var.project
  region  = var.region
  zone    = var.zone
}
