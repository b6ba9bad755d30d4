import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import bench, helpers as H
from naruto_amd import ops, synthetic as syn
gpu = torch.device('cuda:0')
cfg = bench.workload_config("office0_2048x128") if hasattr(bench, "workload_config") else None
