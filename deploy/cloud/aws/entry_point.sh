#!/usr/bin/env bash
# EKS cluster with a CPU node group, a GPU node group and (optionally) an EFS-backed storage
# class for model weights, then the serving chart.
#
#   AWS_REGION=us-west-2 GPU_INSTANCE_TYPE=<AMD Instinct instance type> \
#     deploy/cloud/aws/entry_point.sh setup deploy/cloud/aws/production_stack_specification.yaml
#   deploy/cloud/aws/entry_point.sh status | smoke | cleanup
#
# Environment: CLUSTER_NAME (vpsa-eks), AWS_REGION (us-west-2), GPU_INSTANCE_TYPE (required: no
# default, AMD Instinct instance families differ per account and region), GPU_NODES (1),
# GPU_VOLUME_GB (500), K8S_VERSION (1.30), WITH_EFS (1: run set_up_efs.sh).
# Same flow as the reference's EKS script (/root/reference/deployment_on_cloud/aws/entry_point.sh:1-64),
# with the AMD device plugin in place of the NVIDIA one.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
# shellcheck source=../common/lib.sh
. "$here/../common/lib.sh"

CLUSTER=${CLUSTER_NAME:-vpsa-eks}
REGION=${AWS_REGION:-us-west-2}
GPU_NODES=${GPU_NODES:-1}
GPU_VOLUME_GB=${GPU_VOLUME_GB:-500}
K8S_VERSION=${K8S_VERSION:-1.30}
WITH_EFS=${WITH_EFS:-1}

case "${1:-}" in
  setup)
    values=${2:?usage: $0 setup <values.yaml>}
    GPU_INSTANCE_TYPE=${GPU_INSTANCE_TYPE:?set GPU_INSTANCE_TYPE to an AMD Instinct instance type of your account}
    need aws eksctl kubectl helm
    aws sts get-caller-identity >/dev/null || { echo "aws credentials are not configured" >&2; exit 3; }
    if ! eksctl get cluster --name "$CLUSTER" --region "$REGION" >/dev/null 2>&1; then
      eksctl create cluster --name "$CLUSTER" --region "$REGION" --version "$K8S_VERSION" \
        --nodegroup-name cpu --node-type m6i.2xlarge --nodes 1 --with-oidc
    fi
    if ! eksctl get nodegroup --cluster "$CLUSTER" --region "$REGION" --name gpu >/dev/null 2>&1; then
      eksctl create nodegroup --cluster "$CLUSTER" --region "$REGION" --name gpu \
        --node-type "$GPU_INSTANCE_TYPE" --nodes "$GPU_NODES" --nodes-min "$GPU_NODES" --nodes-max "$GPU_NODES" \
        --node-volume-size "$GPU_VOLUME_GB" --node-labels accelerator=amd-instinct
    fi
    aws eks update-kubeconfig --name "$CLUSTER" --region "$REGION"
    wait_nodes_ready accelerator=amd-instinct "$GPU_NODES"
    install_gpu_plugin
    extra=()
    if [ "$WITH_EFS" = "1" ]; then
      "$here/set_up_efs.sh" "$CLUSTER" "$REGION"
      extra+=(--set "servingEngineSpec.modelSpec[0].storageClass=efs-sc")
    fi
    install_stack "$values" "${extra[@]}"
    ;;
  status)
    kubectl get nodes -L accelerator -o wide
    kubectl -n "$NAMESPACE" get pods,svc,pvc
    ;;
  smoke) smoke_test ;;
  cleanup) exec "$here/clean_up.sh" "$CLUSTER" "$REGION" ;;
  *) echo "usage: $0 setup <values.yaml> | status | smoke | cleanup" >&2; exit 2 ;;
esac
