"""Times the tool-level bundle adjustment alone (bench.py's section).  Usage: python scripts/toolba_bench.py [--cpu]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import cvx_amd  # noqa: E402

print(json.dumps(bench.toolba_bench(cvx_amd, torch, 0, with_cpu="--cpu" in sys.argv), indent=1))
