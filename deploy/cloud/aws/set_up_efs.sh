#!/usr/bin/env bash
# An EFS file system reachable from the cluster's subnets, the EFS CSI driver, and a storage class
# "efs-sc" (ReadWriteMany: every engine replica mounts the same model weights).
#   set_up_efs.sh <cluster> <region>
# Mirrors /root/reference/deployment_on_cloud/aws/set_up_efs.sh:1-73.
set -euo pipefail
CLUSTER=${1:?cluster name}
REGION=${2:?region}

vpc=$(aws eks describe-cluster --name "$CLUSTER" --region "$REGION" --query 'cluster.resourcesVpcConfig.vpcId' --output text)
cidr=$(aws ec2 describe-vpcs --vpc-ids "$vpc" --region "$REGION" --query 'Vpcs[0].CidrBlock' --output text)
echo "cluster VPC $vpc ($cidr)"

sg=$(aws ec2 describe-security-groups --region "$REGION" --filters "Name=group-name,Values=$CLUSTER-efs" "Name=vpc-id,Values=$vpc" \
  --query 'SecurityGroups[0].GroupId' --output text)
if [ "$sg" = "None" ] || [ -z "$sg" ]; then
  sg=$(aws ec2 create-security-group --region "$REGION" --group-name "$CLUSTER-efs" --description "NFS from the $CLUSTER nodes" \
    --vpc-id "$vpc" --query GroupId --output text)
  aws ec2 authorize-security-group-ingress --region "$REGION" --group-id "$sg" --protocol tcp --port 2049 --cidr "$cidr"
fi

fs=$(aws efs describe-file-systems --region "$REGION" --creation-token "$CLUSTER-models" --query 'FileSystems[0].FileSystemId' --output text)
if [ "$fs" = "None" ] || [ -z "$fs" ]; then
  fs=$(aws efs create-file-system --region "$REGION" --creation-token "$CLUSTER-models" --performance-mode generalPurpose \
    --throughput-mode elastic --encrypted --tags "Key=Name,Value=$CLUSTER-models" --query FileSystemId --output text)
fi
echo "file system $fs"
for _ in $(seq 30); do
  state=$(aws efs describe-file-systems --region "$REGION" --file-system-id "$fs" --query 'FileSystems[0].LifeCycleState' --output text)
  [ "$state" = "available" ] && break
  sleep 5
done

subnets=$(aws eks describe-cluster --name "$CLUSTER" --region "$REGION" --query 'cluster.resourcesVpcConfig.subnetIds' --output text)
for sn in $subnets; do
  aws efs create-mount-target --region "$REGION" --file-system-id "$fs" --subnet-id "$sn" --security-groups "$sg" >/dev/null 2>&1 \
    || true  # one mount target per availability zone: the second subnet of a zone is refused, which is fine
done

eksctl create iamserviceaccount --cluster "$CLUSTER" --region "$REGION" --namespace kube-system --name efs-csi-controller-sa \
  --attach-policy-arn arn:aws:iam::aws:policy/service-role/AmazonEFSCSIDriverPolicy --approve --override-existing-serviceaccounts
helm repo add aws-efs-csi-driver https://kubernetes-sigs.github.io/aws-efs-csi-driver/ >/dev/null
helm repo update >/dev/null
helm upgrade --install aws-efs-csi-driver aws-efs-csi-driver/aws-efs-csi-driver --namespace kube-system \
  --set controller.serviceAccount.create=false --set controller.serviceAccount.name=efs-csi-controller-sa

kubectl apply -f - <<YAML
apiVersion: storage.k8s.io/v1
kind: StorageClass
metadata:
  name: efs-sc
provisioner: efs.csi.aws.com
parameters:
  provisioningMode: efs-ap
  fileSystemId: $fs
  directoryPerms: "700"
reclaimPolicy: Delete
volumeBindingMode: Immediate
YAML
echo "storage class efs-sc -> $fs"
