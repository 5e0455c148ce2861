import cProfile, pstats, os, sys, io
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ntsynt_amd import cli, pipeline, synth
d="/tmp/e2e_prof"; os.makedirs(d, exist_ok=True)
paths = synth.make_family(d, 3, 400_000_000, 8, 0.01, micro=20)
os.chdir(d)
pr = cProfile.Profile()
pr.enable()
eng = pipeline.run(paths, k=24, w=1000, prefix="p", w_rounds=[250,100], indel=50000, merge=100000, block_size=1000, benchmark=False, log=lambda *a: None)
pr.disable()
print(eng.stage_times)
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:9000])
