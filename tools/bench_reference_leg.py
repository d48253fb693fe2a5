"""Only the `reference_precision` leg of bench.py (precision="reference" on the headline workload, same runner shape): images/s on stdout.
    python tools/bench_reference_leg.py [bench.py flags]"""
import json
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402

import bench  # noqa: E402
import padt_amd  # noqa: E402

args = bench.parse_args()
args.policy = args.operands
args.operands = "fp16"
torch.cuda.set_device(0)
out = bench.reference_precision_leg(padt_amd.padt_pro_3b(), args, (46, 46), "cuda:0")
out.pop("note", None)
print(json.dumps(out))
