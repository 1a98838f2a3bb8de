#!/usr/bin/env python3
"""Headline benchmark: positive triples/s of the training step (BASELINE.json metric).

Workload (BASELINE.json configs[1], the configuration the metric is quoted on): FB15k-237-shaped
synthetic knowledge graph (14 541 entities, 237 relations, 272 115 train triples; the reference
ships no data), TransE dim=200, Adam with the reference's TF1 semantics (dense sweep), 25
negatives per positive with Bernoulli head/tail skew, margin 1.0.  Arithmetic and storage are fp32
(the parity mode; BASELINE's "bf16" storage would be a precision reduction and is not used here).

One "step" = what the reference does per loop iteration (distribute_training.py:274-282):
sample a batch (on the device), forward, backward, optimiser update -- all inside the timed region,
inputs resident in HBM.  N GPUs = N processes, rank g owns the virtual sampler threads
[g*8/N, (g+1)*8/N); the integer count image is reduce-scattered over RCCL, every rank applies Adam to its
chunk of the tables and the updated chunks are all-gathered.  Per-GPU batch is held at ~34 014 positives
(nbatches = 8/N), so scaling is weak.

python bench.py --gpus N --steps K --warmup W
N > 1 from a bare shell: this process starts the N ranks itself (`python -m torch.distributed.run`, one rank per GPU, RCCL over
xGMI) BEFORE anything touches a GPU, relays rank 0's JSON line and exits with the job's status -- what the reference's launcher
does with TFCluster.run (main_spark.py:340).  Under an external torchrun (WORLD_SIZE set) it is one of the ranks.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

DIM = 200
NEG = 25
WORK_THREADS = 8
PER_GPU_NBATCHES = 8     # nbatches = 8/N  -> per-GPU batch 34 014, global batch 34 014 * N
HBM_PEAK_GBS = 8000.0    # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)


def algorithmic_bytes_per_positive(n_neg, dim, elem=4):
    """SURVEY.md 8(d), TransE: U = 3+n unique rows per positive group.  The dominant kernel (the
    TransE emit kernel: gather -> normalise -> score -> hinge -> int8 sign records) is priced at the
    survey's GATHER figure U*D*s plus the int32 (h,t,r) of the 1+n scored triples; the 256-byte int8
    gradient records it also writes (U per positive) are NOT counted, so this is the conservative
    number."""
    u = 3 + n_neg
    return u * dim * elem + 12 * (1 + n_neg)


def usable_cpus(omp_max):
    """CPUs this job may actually use: the smaller of OpenMP's count, the affinity mask and the cgroup
    quota (a 1-GPU box grants a share of the host, and oversubscribing it only measures contention)."""
    n = omp_max
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("KGE_CPU_BASELINE_THREADS", "64"))))


def cpu_baseline(fb_dir, B, seconds=15.0):
    """The CPU oracle (C restatement of the reference path: sampler + TransE fwd/bwd + TF1 Adam) timed on this host on
    a bounded sample of the SAME workload (same graph, same per-step batch).  The fast CPU form: forward/backward over all
    granted cores with thread-private gradient accumulators (no atomic adds), the sampler with one OS thread per virtual
    thread as the reference's pthreads run it (Base.cpp:151-171) or serially, whichever is quicker here."""
    from oracle import oracle
    threads = usable_cpus(oracle.lib().orc_max_threads())
    kg = oracle.KG(fb_dir, work_threads=WORK_THREADS, bern=1)
    m = oracle.Model("transe", kg.entTotal, kg.relTotal, DIM, margin=1.0, seed=0)
    timings = {}
    for par in (False, True):     # warm-up doubles as the sampler choice
        t0 = time.perf_counter()
        bh, bt, br, _ = kg.sampling(B, NEG, 0, parallel=par)
        timings[par] = time.perf_counter() - t0
    par = timings[True] < timings[False]
    m.adam_step(bh, bt, br, B, NEG, 0.001, nthreads=-threads if threads > 1 else 1)
    t0 = time.perf_counter()
    steps = 0
    while True:
        bh, bt, br, _ = kg.sampling(B, NEG, 0, parallel=par)
        m.adam_step(bh, bt, br, B, NEG, 0.001, nthreads=-threads if threads > 1 else 1)
        steps += 1
        if time.perf_counter() - t0 >= seconds or steps >= 200:
            break
    dt = time.perf_counter() - t0
    return {"value": B * steps / dt, "unit": "positive triples/s", "cores": int(threads), "kind": "port",
            "sample": "%d steps of B=%d positives x %d negatives (the GPU's per-step batch), TransE dim=%d, TF1 Adam; "
                      "oracle/kge_oracle.c: OpenMP forward/backward with thread-private accumulators, sampler %s"
                      % (steps, B, NEG, DIM, "one thread per virtual thread" if par else "single thread")}


def cpu_model_string():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline_config1(fb_dir, warm=100, timed=1000):
    """SURVEY.md 8(d) / BASELINE.md 3, to the letter: BASELINE config #1 -- FB15k-237(-shaped) TransE dim 100, L1, margin 1.0,
    SGD, 1 negative per positive, UNIFORM corruption, the reference's auto batch B = 2 721 (Config.py:189-210), fp32 -- timed
    on this host at workThreads = 1 and at workThreads = all granted cores; >= 100 warm steps, >= 1 000 timed steps, each
    step timed on its own: median and p10 / p90 of positives/s = B / step time.  A step = `sampling` (Base.cpp:149-172) +
    forward / backward / scatter_sub (TransE.py:26-51, distribute_training.py:98-101) as the oracle restates them; the TF1
    graph executor's own overhead is NOT in it, so the real reference is slower than this."""
    import numpy as np
    from oracle import oracle
    cores = usable_cpus(oracle.lib().orc_max_threads())
    legs = {}
    for name, w in (("workThreads_1", 1), ("workThreads_all", cores)):
        kg = oracle.KG(fb_dir, work_threads=w, bern=0)
        m = oracle.Model("transe", kg.entTotal, kg.relTotal, 100, margin=1.0, seed=0)
        B = 2721
        par, nth, form = False, 1, "one thread"
        if w > 1:
            # the faster of the CPU forms at this batch, chosen in the warm-up (keeps the speed-up claim conservative): sampler
            # with pthread-style slices or serial; gradient rows added with atomics or into thread-private images
            best = None
            for par_c in (False, True):
                for nth_c, form_c in ((w, "shared accumulators, atomic adds"), (-w, "thread-private accumulators")):
                    t0 = time.perf_counter()
                    for _ in range(20):
                        bh, bt, br, _y = kg.sampling(B, 1, 0, parallel=par_c)
                        m.sgd_step(bh, bt, br, B, 1, 0.01, nthreads=nth_c)
                    dt_c = time.perf_counter() - t0
                    if best is None or dt_c < best[0]:
                        best = (dt_c, par_c, nth_c, form_c)
            _, par, nth, form = best
            form += ", sampler %s" % ("one OS thread per virtual thread" if par else "serial over the virtual threads")
        for _ in range(warm):
            bh, bt, br, _y = kg.sampling(B, 1, 0, parallel=par)
            m.sgd_step(bh, bt, br, B, 1, 0.01, nthreads=nth)
        t = np.zeros(timed)
        for i in range(timed):
            t0 = time.perf_counter()
            bh, bt, br, _y = kg.sampling(B, 1, 0, parallel=par)
            m.sgd_step(bh, bt, br, B, 1, 0.01, nthreads=nth)
            t[i] = time.perf_counter() - t0
        rate = B / t
        legs[name] = {"threads": int(w), "median": float(np.median(rate)), "p10": float(np.percentile(rate, 10)),
                      "p90": float(np.percentile(rate, 90)), "warm_steps": warm, "timed_steps": timed, "form": form,
                      "ms_per_step_median": float(np.median(t) * 1e3)}
    return {"unit": "positive triples/s", "kind": "port", "cpu_model": cpu_model_string(), "cores_granted": int(cores),
            "config": "BASELINE configs[0]: FB15k-237-shaped TransE dim=100 L1 margin=1.0 SGD lr 0.01, 1 neg/pos uniform, B=2721 "
                      "(auto batch, 100 batches/epoch), fp32",
            "protocol": "SURVEY.md 8(d): per-step wall time of sampling + forward/backward/SGD, >=100 warm + >=1000 timed steps, "
                        "median and p10/p90 of B/step_time; TF1 executor overhead not included",
            **legs}


def gpu_config1(fb_dir, device, steps=2000):
    """The engine on the same configuration #1, for the line beside cpu_baseline_config1: separate launches per step
    (Config.train_step) and the persistent many-steps-per-launch form (Config.train_steps, csrc/persist.hip)."""
    import torch
    from openkeonspark_amd import Config, TransE
    out = {}
    for name, persistent in (("separate_launches", False), ("persistent_launch", True)):
        con = Config()
        con.device = device
        con.set_in_path(fb_dir); con.set_work_threads(WORK_THREADS); con.set_bern(0); con.set_dimension(100); con.set_nbatches(0)
        con.set_ent_neg_rate(1); con.set_rel_neg_rate(0); con.set_margin(1.0); con.set_alpha(0.01); con.set_opt_method("SGD")
        con.init()
        con.set_model_and_session(TransE)      # (defaults: the next batch's sampler rides in the forward/backward launch)
        assert con.batch_size == 2721
        if persistent and not con.persistent_supported():
            continue
        con.train_steps(100, persistent=persistent)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        con.train_steps(steps, persistent=persistent)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[name] = {"value": 2721 * steps / dt, "us_per_step": dt / steps * 1e6, "steps": steps}
    return out


def exchange_rehearsal(steps=100, warmup=20, timeout=240):
    """The same workload once more through the data-parallel exchange, as far as one GPU allows: a CHILD process runs this
    script on a one-rank RCCL group with Config.force_data_parallel (every collective of the step is issued; with one rank each
    is a copy).  Reported beside the headline so that the N = 1 line already shows what the exchange machinery costs before any
    byte crosses xGMI; not part of `value`.  Any failure is reported, never raised: it must not take the headline down."""
    import subprocess
    env = dict(os.environ, KGE_BENCH_FORCE_DIST="1", KGE_BENCH_FORCE_DP="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(29650 + os.getpid() % 300), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    try:
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--steps", str(steps), "--warmup", str(warmup), "--no-cpu-baseline"],
                           env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=timeout, check=False)
        line = [l for l in p.stdout.decode(errors="replace").splitlines() if l.startswith("{")][-1]
        d = json.loads(line)
        return {"what": "one-rank RCCL group, force_data_parallel: reduce-scatter / owned-share Adam / all-gather issued every step",
                "ms_per_step": d["ms_per_step"], "value": d["value"], "backend": d["backend"], "rccl_ranks": d["rccl_ranks"],
                "collective_stream": d.get("collective_stream"), "steps": d["steps"], "warmup": d["warmup"]}
    except Exception as exc:      # noqa: BLE001 -- reported in the line instead
        return {"error": "%s: %s" % (type(exc).__name__, str(exc)[:200])}


def launch_ranks(n, argv):
    """Start `n` ranks of this script on this node and relay rank 0's line.  Runs in a process that has not initialised the GPU
    (no HIP call, no torch.cuda call before this point), starts the ranks as CHILD processes and returns their status: nothing is
    exec'ed over a process that holds a device."""
    import socket
    import subprocess
    with socket.socket() as sock:                       # a free rendezvous port on the loopback interface
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this stack
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=sys.stderr)
    line = None
    for raw in proc.stdout:                              # only rank 0's JSON line goes to our stdout
        text = raw.decode(errors="replace").rstrip("\n")
        if text.startswith("{"):
            line = text
        elif text:
            print(text, file=sys.stderr)
    rc = proc.wait()
    if rc != 0:
        print("bench.py: the %d-rank job failed (exit status %d)" % (n, rc), file=sys.stderr)
        return rc if rc > 0 else 1
    if line is None:
        print("bench.py: the %d-rank job printed no result line" % n, file=sys.stderr)
        return 1
    print(line)
    sys.stdout.flush()
    return 0


def eval_leg(device, epochs=6):
    """The second half of BASELINE's metric -- filtered MR / Hits@10 -- on a bounded run (tests/test_gpu_metric.py's protocol,
    shortened): a typed synthetic graph with FB15k-237's cardinalities (learnable structure; the reference ships no data), TransE
    dim 100, SGD, 4 negatives, the reference's auto batch 2 721; the ENGINE trains `epochs` x 100 steps on device-sampled
    batches and the CPU ORACLE the same steps on the bit-identical batches from the same initial tables; both resulting models
    are ranked over the whole test split (20 466 triples x 14 541 candidates x 2 sides, raw + filtered) by the device ranker
    (csrc/eval.hip, bit-exact against the compiled reference's testHead / testTail: tests/test_gpu_lp.py), which is also timed.
    Reference: base/Test.h:31-249, distribute_training.py:465-527, main_spark.py:430-448."""
    import numpy as np
    import torch
    from openkeonspark_amd import Config, TransE
    from openkeonspark_amd.synthetic import make_typed_dataset, FB15K237_TYPED
    from oracle import oracle
    path = make_typed_dataset("/tmp/okes_typed_fb", FB15K237_TYPED)
    dim, n, alpha = 100, 4, 10.0
    con = Config()
    con.device = device
    con.prefetch_sampling = False
    con.set_in_path(path); con.set_work_threads(WORK_THREADS); con.set_bern(0); con.set_dimension(dim); con.set_nbatches(100)
    con.set_ent_neg_rate(n); con.set_alpha(alpha); con.set_margin(1.0); con.set_opt_method("SGD")
    con.set_test_link_prediction(True)
    con.init()
    con.set_model_and_session(TransE)
    kg = oracle.KG(path, work_threads=WORK_THREADS, bern=0)
    kg.set_stream_states(con.get_stream_states())
    orc = oracle.Model("transe", con.entTotal, con.relTotal, dim, dim, margin=1.0, params=con.get_parameters())
    B, steps = con.batch_size, epochs * con.nbatches
    threads = usable_cpus(oracle.lib().orc_max_threads())
    _, untrained = con.link_prediction(test_head=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        con.train_step(sync=False)
    torch.cuda.synchronize()
    t_engine = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(steps):
        bh, bt, br, _y = kg.sampling(B, n, 0)
        orc.sgd_step(bh, bt, br, B, n, alpha, nthreads=min(threads, 8))
    t_oracle = time.perf_counter() - t0
    same_batches = con.get_stream_states().tolist() == kg.stream_states().tolist()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out_g, met_g = con.link_prediction(test_head=True)
    torch.cuda.synchronize()
    t_rank = time.perf_counter() - t0
    con.set_parameters(orc.params)
    _, met_o = con.link_prediction(test_head=True)
    n_test = int(out_g.shape[0])
    pick = lambda m: {"MR_filter_tail": m["r_filter_rank"], "MR_filter_head": m["l_filter_rank"], "Hits10_filter_tail": m["r_filter_tot"],
                      "Hits10_filter_head": m["l_filter_tot"], "MR_raw_tail": m["r_rank"], "Hits10_raw_tail": m["r_tot"]}
    scores = 2.0 * n_test * con.entTotal
    return {"what": "typed synthetic KG with FB15k-237's cardinalities, TransE dim %d, SGD lr %g, %d neg/pos, B = %d, %d steps; engine-trained "
                    "vs oracle-trained (same batches, same initial tables) ranked over the whole test split by the device ranker" % (dim, alpha, n, B, steps),
            "test_triples": n_test, "candidates_per_side": int(con.entTotal), "same_batches_drawn": bool(same_batches),
            "untrained": pick(untrained), "engine_trained": pick(met_g), "oracle_trained": pick(met_o),
            "train_seconds": {"engine": t_engine, "oracle_cpu": t_oracle, "oracle_threads": int(min(threads, 8))},
            "ranker": {"seconds": t_rank, "test_triples_per_s": n_test / t_rank, "candidate_scores_per_s": scores / t_rank,
                       "achieved_GBps": scores * dim * 4 / t_rank / 1e9,
                       "roofline_note": "each candidate score reads one %d-float entity row (the 5.8 MB table is cache-resident: a gather "
                                        "from L2 / Infinity Cache, not HBM) plus the filter's binary searches; counted: scores x dim x 4 B" % dim}}


def adam_step_bytes(ent_total, rel_total, dim):
    """TF1 'sparse' Adam is a dense sweep: every element of p, m, v is read and written each step
    (24 B) and the summed gradient image is read and re-zeroed (8 B)."""
    return (ent_total + rel_total) * dim * 32


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: long enough to reach the steady state of this workload -- a random-init model starts with every hinge active, so
    # the first ~30 steps carry more gradient records than the rest of an epoch (0.27 ms/step over steps 5..25, 0.22 from step 20 on)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--prefetch-sampling", choices=["auto", "on", "off"], default="auto",
                    help="batch i+1 sampled on a side stream behind the emit kernel of step i (Config.prefetch_sampling; "
                         "bit-identical batches).  auto = the Config default")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        if PER_GPU_NBATCHES % args.gpus:
            raise SystemExit("--gpus must divide %d" % PER_GPU_NBATCHES)
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))     # (before any GPU call in this process)

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world       # an external launcher's world size wins over the flag
    if os.environ.get("KGE_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0   # rehearsal of the N-rank code path on a one-GPU box (with KGE_BENCH_BACKEND=gloo)
    torch.cuda.set_device(local_rank)
    # KGE_BENCH_FORCE_DIST=1 initialises the RCCL process group even with a single rank
    use_dist = world > 1 or os.environ.get("KGE_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        backend = os.environ.get("KGE_BENCH_BACKEND", "nccl")   # "nccl" IS RCCL on ROCm; gloo only for one-GPU rehearsals
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from openkeonspark_amd.synthetic import make_dataset, FB15K237
    from openkeonspark_amd import Config, TransE

    if rank == 0:
        fb_dir = make_dataset("/tmp/okes_fb15k237_shaped", FB15K237)
    if use_dist:
        dist.barrier()
    fb_dir = make_dataset("/tmp/okes_fb15k237_shaped", FB15K237)

    if PER_GPU_NBATCHES % world:
        raise SystemExit("--gpus must divide %d" % PER_GPU_NBATCHES)
    # the reference-style prints (Python "Batch size is ..." and the C library's "Input Files Path
    # : ...") must not pollute the single JSON line: park fd 1 on /dev/null while setting up
    sys.stdout.flush()
    saved_fd = os.dup(1)
    devnull_fd = os.open(os.devnull, os.O_WRONLY)
    os.dup2(devnull_fd, 1)
    con = Config()
    for key, val in os.environ.items():      # KGE_OPT_<engine option>=<int>: A/B runs of engine variants (include/kge_mi355.h kge_set_option)
        if key.startswith("KGE_OPT_"):
            con.lib.kge_set_option(key[8:].lower().encode(), int(val))
    con.device = "cuda:%d" % local_rank
    con.set_in_path(fb_dir)
    con.set_work_threads(WORK_THREADS)
    con.set_bern(1)
    con.set_dimension(DIM)
    con.set_nbatches(PER_GPU_NBATCHES // world)
    con.set_ent_neg_rate(NEG)
    con.set_rel_neg_rate(0)
    con.set_margin(1.0)
    con.set_alpha(0.001)
    con.set_opt_method("Adam")
    if args.prefetch_sampling != "auto":
        con.prefetch_sampling = args.prefetch_sampling == "on"
    con.init()
    con.set_model_and_session(TransE)
    if use_dist:
        if os.environ.get("KGE_BENCH_FORCE_DP") == "1":     # one-GPU rehearsal: a one-rank group takes the whole data-parallel exchange
            con.force_data_parallel = True
        con.init_distributed()
    sys.stdout.flush()
    con.lib.kge_clear_error()
    import ctypes
    ctypes.CDLL(None).fflush(None)
    os.dup2(saved_fd, 1)
    os.close(saved_fd)
    os.close(devnull_fd)
    B = con.batch_size
    n_local = con._n_local

    def sync():
        torch.cuda.synchronize()      # (drained before the process group's barrier: the step's collectives may be on the engine's own communicator)
        if use_dist:
            con.comm_fence("pg")
            dist.barrier()
        torch.cuda.synchronize()

    import ctypes
    from openkeonspark_amd import _lib as _l
    for _ in range(args.warmup):
        con.train_step(sync=False)
    sync()
    # roofline of the dominant kernel (TransE emit): one HIP event pair per sampled launch, recorded on the kernel's launch stream
    # INSIDE the timed region and read back after it (no synchronisation between the steps)
    con.lib.kge_set_option(b"time_emit", 4)   # every 4th launch: the event records themselves cost ~2 % when on every step
    t0 = time.perf_counter()
    for _ in range(args.steps):
        con.train_step(sync=False)
    sync()
    dt = time.perf_counter() - t0
    if use_dist:
        con.comm_fence("pg")
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss = float(con._loss.item())
    # how many ranks the collective library really joined: an all-reduce of ones over the backend the step uses
    backend_name, rccl_ranks = "none (single process, no process group)", 1
    if use_dist:
        ones = torch.ones(1, device="cuda", dtype=torch.float32)
        dist.all_reduce(ones)
        rccl_ranks = int(round(float(ones.item())))
        backend_name = dist.get_backend()
        if backend_name == "nccl":
            backend_name = "nccl (RCCL on ROCm)"
    # where the step's reduce-scatter / all-gather were enqueued (openkeonspark_amd/parallel.py StreamRccl)
    collective_stream = "none" if not getattr(con, "_dp", False) else (
        "the engine's stream (own RCCL communicator, %d ranks)" % con._nat_rccl.world if getattr(con, "_nat_rccl", None) else "the process group's stream")
    # the kernel-duration sample must not depend on how few steps the caller timed: keep stepping (outside the timed
    # region, same training run) until at least 50 launches carry an event pair
    ms = ctypes.c_float()
    timed = ctypes.c_int64()
    in_region = (args.steps + 3) // 4
    extra = 0
    # --steps >= 200: all >= 50 timed launches lie INSIDE the timed region and nothing more is run; shorter runs top the sample up
    # with further steps of the same training run after the region (the regime is reported: roofline.kernel_ms_regime)
    while in_region + extra // 4 < 50 and extra < 400:
        con.train_step(sync=False)
        extra += 1
    sync()
    con.lib.kge_set_option(b"time_emit", 0)
    _l.check(con.lib.kge_kernel_ms_mean(b"transe_emit", ctypes.byref(ms), ctypes.byref(timed)), con.lib)
    kern_ms = float(ms.value)
    n_pos = n_local
    alg_bytes = algorithmic_bytes_per_positive(NEG, DIM) * n_pos
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
    traffic, traffic_note = None, None
    tr_path = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tr_path):
        try:
            tr = json.load(open(tr_path))
            traffic = tr.get("emit_hbm_bytes_per_launch")
            traffic_note = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command at commit %s (%s)" % (
                tr.get("measured_at_commit", "?"), tr.get("source", "profiles/"))
        except Exception:
            traffic = None
    # whole-step figure: the gather the emit kernel does + TF1 Adam's dense sweep, over the step time (all kernels)
    step_bytes = alg_bytes + adam_step_bytes(con.entTotal, con.relTotal, DIM)
    step_gbps = step_bytes / (dt / args.steps) / 1e9

    if rank == 0:
        out = {
            "metric": "positive triples/sec (training step)",
            "value": B * args.steps / dt,
            "unit": "positive triples/s",
            "n_gpus": world, "world_size": dist.get_world_size() if use_dist else 1, "backend": backend_name,
            "rccl_ranks": rccl_ranks, "collective_stream": collective_stream, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            # which part of a training run the timed window covers: a random-init model starts with every hinge active (all 25 negatives
            # of a group contribute gradient records); from ~step 25 on about a third are (steady state, what an epoch mostly consists of)
            "regime": ("steps %d..%d of a random-init run: %s" % (args.warmup, args.warmup + args.steps - 1,
                       "all-hinges-active regime (the first ~25 steps carry ~3x the gradient records of the steady state)"
                       if args.warmup + args.steps <= 40 else "mostly steady state (from ~step 25 on a third of the hinges are active)")),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "FB15k-237-shaped synthetic KG (E=14541,R=237,272115 triples), TransE dim=200, "
                                   "TF1-semantics Adam, 25 neg/pos bern, margin 1.0 (BASELINE configs[1] in fp32, the reference's "
                                   "precision: TransE.py:21-22)",
                       "global_batch": B, "per_gpu_batch": n_local, "neg_per_pos": NEG, "dim": DIM,
                       "optimizer": "Adam(dense, TF1 parity)", "gradient_path": "int8 sign-count records (exact)",
                       "work_threads": WORK_THREADS,
                       "parallelism": "dp%d" % world, "final_loss": loss},
            "roofline": {"bound": "hbm", "kernel": "kge::transe_emit_vec_kernel<64,1,4,1,true,true>",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_note, "kernel_ms": kern_ms,
                         "kernel_launches_timed": int(timed.value), "kernel_launches_timed_inside_region": min(in_region, int(timed.value)),
                         "kernel_ms_regime": ("all timed launches inside the timed region" if extra == 0 else
                                              "%d launches inside the timed region + further steps of the same run after it "
                                              "(--steps >= 200 keeps all of them inside)" % in_region),
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "step": {"algorithmic_bytes": step_bytes, "achieved": step_gbps, "frac": step_gbps / HBM_PEAK_GBS,
                                  "unit": "GB/s", "what": "gather bytes of the step + the dense TF1 Adam sweep, over ms_per_step"}},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(fb_dir, n_local)
            # the survey's own protocol (config #1, workThreads 1 and all cores, median + p10/p90) with the engine beside it
            sys.stdout.flush()
            saved_fd = os.dup(1)
            devnull_fd = os.open(os.devnull, os.O_WRONLY)
            os.dup2(devnull_fd, 1)
            try:
                c1 = cpu_baseline_config1(fb_dir)
                c1["gpu_same_config"] = gpu_config1(fb_dir, "cuda:%d" % local_rank)
            finally:
                sys.stdout.flush()
                ctypes.CDLL(None).fflush(None)
                os.dup2(saved_fd, 1)
                os.close(saved_fd)
                os.close(devnull_fd)
            out["cpu_baseline_config1"] = c1
            try:
                sys.stdout.flush()
                saved_fd = os.dup(1)
                devnull_fd = os.open(os.devnull, os.O_WRONLY)
                os.dup2(devnull_fd, 1)
                try:
                    out["eval"] = eval_leg("cuda:%d" % local_rank)
                finally:
                    sys.stdout.flush()
                    ctypes.CDLL(None).fflush(None)
                    os.dup2(saved_fd, 1)
                    os.close(saved_fd)
                    os.close(devnull_fd)
            except Exception as exc:      # noqa: BLE001 -- reported in the line, never takes the headline down
                out["eval"] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:200])}
            if not use_dist:
                out["exchange_rehearsal"] = exchange_rehearsal()
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
