# Local state by default; see ../gke-infrastructure/backend.tf for a bucket backend.
