#!/usr/bin/env bash
# Remove everything entry_point.sh and set_up_efs.sh created, in the order AWS allows:
# release and volumes -> load balancers -> EFS mount targets, file system, security group ->
# node groups and cluster.      clean_up.sh <cluster> <region>
# (/root/reference/deployment_on_cloud/aws/clean_up.sh:1-117 walks the same dependency chain.)
set -uo pipefail
CLUSTER=${1:?cluster name}
REGION=${2:?region}
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
. "$here/../common/lib.sh"

if aws eks update-kubeconfig --name "$CLUSTER" --region "$REGION" >/dev/null 2>&1; then
  uninstall_stack
  helm uninstall aws-efs-csi-driver --namespace kube-system 2>/dev/null || true
  kubectl delete storageclass efs-sc --ignore-not-found
fi

fs=$(aws efs describe-file-systems --region "$REGION" --creation-token "$CLUSTER-models" --query 'FileSystems[0].FileSystemId' --output text 2>/dev/null)
if [ -n "$fs" ] && [ "$fs" != "None" ]; then
  for mt in $(aws efs describe-mount-targets --region "$REGION" --file-system-id "$fs" --query 'MountTargets[].MountTargetId' --output text); do
    aws efs delete-mount-target --region "$REGION" --mount-target-id "$mt"
  done
  for _ in $(seq 30); do  # the file system cannot go while mount targets are still deleting
    left=$(aws efs describe-mount-targets --region "$REGION" --file-system-id "$fs" --query 'length(MountTargets)' --output text)
    [ "$left" = "0" ] && break
    sleep 5
  done
  aws efs delete-file-system --region "$REGION" --file-system-id "$fs" && echo "deleted $fs"
fi

vpc=$(aws eks describe-cluster --name "$CLUSTER" --region "$REGION" --query 'cluster.resourcesVpcConfig.vpcId' --output text 2>/dev/null)
if [ -n "$vpc" ] && [ "$vpc" != "None" ]; then
  sg=$(aws ec2 describe-security-groups --region "$REGION" --filters "Name=group-name,Values=$CLUSTER-efs" "Name=vpc-id,Values=$vpc" \
    --query 'SecurityGroups[0].GroupId' --output text)
  [ -n "$sg" ] && [ "$sg" != "None" ] && aws ec2 delete-security-group --region "$REGION" --group-id "$sg"
fi

eksctl delete iamserviceaccount --cluster "$CLUSTER" --region "$REGION" --namespace kube-system --name efs-csi-controller-sa 2>/dev/null || true
eksctl delete cluster --name "$CLUSTER" --region "$REGION" --wait
echo "left behind on purpose: CloudWatch log groups /aws/eks/$CLUSTER/* (delete them if you do not want the logs)"
