#!/usr/bin/env bash
# CRI-O as the container runtime of a kubeadm node (the alternative to the containerd that
# utils/install-kubeadm.sh sets up).  Run BEFORE kubeadm init / join and pass
# --cri-socket unix:///var/run/crio/crio.sock to kubeadm.
set -euo pipefail
CRIO_MINOR=${CRIO_MINOR:-v1.31}
sudo apt-get update && sudo apt-get install -y software-properties-common curl gpg
sudo mkdir -p /etc/apt/keyrings
curl -fsSL "https://pkgs.k8s.io/addons:/cri-o:/stable:/${CRIO_MINOR}/deb/Release.key" |
  sudo gpg --dearmor -o /etc/apt/keyrings/cri-o.gpg
echo "deb [signed-by=/etc/apt/keyrings/cri-o.gpg] https://pkgs.k8s.io/addons:/cri-o:/stable:/${CRIO_MINOR}/deb/ /" |
  sudo tee /etc/apt/sources.list.d/cri-o.list
sudo apt-get update && sudo apt-get install -y cri-o
# kernel prerequisites of any kubeadm node
printf 'overlay\nbr_netfilter\n' | sudo tee /etc/modules-load.d/k8s.conf >/dev/null
sudo modprobe overlay && sudo modprobe br_netfilter
printf 'net.bridge.bridge-nf-call-iptables=1\nnet.ipv4.ip_forward=1\nnet.bridge.bridge-nf-call-ip6tables=1\n' |
  sudo tee /etc/sysctl.d/k8s.conf >/dev/null
sudo sysctl --system >/dev/null
sudo systemctl enable --now crio
sudo swapoff -a
echo "next: sudo kubeadm init --cri-socket unix:///var/run/crio/crio.sock --pod-network-cidr=192.168.0.0/16"
