# State stays local by default.  For a team, put it in a bucket:
# terraform {
#   backend "gcs" {
#     bucket = "<your-tf-state-bucket>"
#     prefix = "vpsa/gke-infrastructure"
#   }
# }
