#!/usr/bin/env python
"""Host-side profile of the drop-in autograd path: cProfile over bench.py --api autograd, printing the entries of this
repository's modules and torch's autograd entry points by cumulative time."""
import cProfile
import io
import pstats
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

sys.argv = ["bench.py", "--no-cpu-baseline", "--api", "autograd", "--steps", "10", "--warmup", "2"] + sys.argv[1:]
pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
out = io.StringIO()
st = pstats.Stats(pr, stream=out).sort_stats("cumulative")
st.print_stats(r"luciddreamer_amd|depth_diff|bench.py|run_backward|apply|torch.empty|torch.zeros|parallel")
print(out.getvalue()[:6000])
