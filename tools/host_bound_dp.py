"""Is the data-parallel step host-bound?  One-rank RCCL group, force_data_parallel: host enqueue time per step vs wall time per step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import openkeonspark_amd as pkg
from openkeonspark_amd.synthetic import make_dataset, FB15K237
con = pkg.Config()
con.set_in_path(make_dataset("/tmp/okes_fb15k237_shaped", dict(FB15K237, name="fb15k237_shaped"))); con.set_work_threads(8); con.set_bern(1)
con.set_dimension(200); con.set_nbatches(8); con.set_ent_neg_rate(25); con.set_alpha(0.001); con.set_opt_method("Adam")
con.init(); con.set_model_and_session(pkg.TransE)
con.force_data_parallel = True
con.init_distributed()
for _ in range(30):
    con.train_step(sync=False)
torch.cuda.synchronize()
N = 200
t0 = time.perf_counter()
for _ in range(N):
    con.train_step(sync=False)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue %.1f us/step, wall %.1f us/step" % (1e6 * (t1 - t0) / N, 1e6 * (t2 - t0) / N))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(100):
    con.train_step(sync=False)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
dist.destroy_process_group()
