"""Host-side profile of the training bench loop (cProfile, cumulative top-N) -- where Python/launch time goes."""
import cProfile
import pstats
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py", "--mode", "train", "--steps", "10", "--warmup", "4"]
import bench  # noqa: E402

pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(70)
