#!/usr/bin/env python3
"""joint perft from the dual start position on the GPU (tools/benchmark.cc:59-97 counting convention)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hivemind_amd as hm
hm.init(0)
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 5
n, secs = hm.perft(depth)
print(json.dumps(dict(depth=depth, nodes=int(n), seconds=secs, nodes_per_s=n / secs if secs > 0 else None)))
