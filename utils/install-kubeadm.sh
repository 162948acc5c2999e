#!/usr/bin/env bash
# kubeadm/kubelet from the upstream apt repository plus containerd, for a bare-metal MI355X node.
set -euo pipefail
K8S_MINOR=${K8S_MINOR:-v1.31}
sudo apt-get update && sudo apt-get install -y apt-transport-https ca-certificates curl gpg containerd
sudo mkdir -p /etc/apt/keyrings /etc/containerd
curl -fsSL "https://pkgs.k8s.io/core:/stable:/${K8S_MINOR}/deb/Release.key" | sudo gpg --dearmor -o /etc/apt/keyrings/kubernetes.gpg
echo "deb [signed-by=/etc/apt/keyrings/kubernetes.gpg] https://pkgs.k8s.io/core:/stable:/${K8S_MINOR}/deb/ /" | sudo tee /etc/apt/sources.list.d/kubernetes.list
sudo apt-get update && sudo apt-get install -y kubelet kubeadm kubectl && sudo apt-mark hold kubelet kubeadm kubectl
containerd config default | sed 's/SystemdCgroup = false/SystemdCgroup = true/' | sudo tee /etc/containerd/config.toml >/dev/null
sudo systemctl restart containerd
sudo swapoff -a
echo "next: sudo kubeadm init --pod-network-cidr=192.168.0.0/16 && utils/install-calico.sh"
