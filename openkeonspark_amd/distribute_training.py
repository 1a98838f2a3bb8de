"""Training / evaluation driver: the loop of /root/reference/distribute_training.py without Spark or TF.

Same flags as `main_spark.py:299-324`, same `get_conf` (`distribute_training.py:32-71`), same per-step
log line (`:283`), same checkpoint discovery (`get_last_step`, `:134-156`: the step is parsed from the
`checkpoint` text file, `model.ckpt-<step>`), one checkpoint per epoch with `max_to_keep =
patience + 5` (`:103,226-234`), loss-based early stop with `stop.txt` (`:336-360`), new-entity growth
on restore (`main_spark.py:74-98`: xavier rows for parameters, zero rows for the Adam slots).  One
process per GPU (torchrun) replaces the ps/worker cluster; rank 0 plays the chief.

    python -m openkeonspark_amd.distribute_training --input_path DATA/ --output_path OUT/ --model TransE ...
    torchrun --nproc-per-node 8 -m openkeonspark_amd.distribute_training ...

Early stop: both criteria of the reference -- validation accuracy of triple classification
(`:295-333`: thresholds from `getBestThreshold` on the validation positives / type-constrained
negatives of `getValidBatch`) when valid2id.txt / test2id.txt / type_constrain.txt are present, and the
loss (`:336-360`).  One deliberate difference: the reference then calls
`test_triple_classification(relThresh, valid_scores...)`, which walks the TEST set's per-relation
ranges over the validation score arrays (`Test.h:352-353` with `distribute_training.py:312`); here the
accuracy is computed over the validation triples the thresholds were fitted on.
"""
import argparse
import glob
import json
import os
import sys
import time

import numpy as np

from .Config import Config
from .Model import xavier_normal
from .TransD import TransD
from .TransE import TransE
from .TransH import TransH
from .TransR import TransR


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="OpenKEonSpark training loop on MI355X")
    p.add_argument("--cluster_size", type=int, default=1, help="accepted for compatibility (ranks come from torchrun)")
    p.add_argument("--num_ps", type=int, default=0, help="accepted for compatibility (no parameter servers)")
    p.add_argument("--num_gpus", type=int, default=1)
    p.add_argument("--cpp_lib_path", type=str, default=None)
    p.add_argument("--input_path", type=str, default=None)
    p.add_argument("--output_path", type=str, default=None)
    p.add_argument("--train_times", type=int, default=100)
    p.add_argument("--n_mini_batches", type=int, default=0)
    p.add_argument("--alpha", type=float, default=0.00001)
    p.add_argument("--margin", type=float, default=1.0)
    p.add_argument("--bern_flag", type=int, default=0)
    p.add_argument("--embedding_dimension", type=int, default=64)
    p.add_argument("--ent_dimension", type=int, default=0)
    p.add_argument("--rel_dimension", type=int, default=0)
    p.add_argument("--ent_neg_rate", type=int, default=1)
    p.add_argument("--rel_neg_rate", type=int, default=0)
    p.add_argument("--optimizer", type=str, default="SGD")
    p.add_argument("--early_stop_patience", type=int, default=5)
    p.add_argument("--early_stop_stopping_step", type=int, default=1)
    p.add_argument("--early_stop_start_step", type=int, default=1)
    p.add_argument("--model", type=str, default="TransE")
    p.add_argument("--debug", type=int, default=0)
    p.add_argument("--mode", type=str, default="train")
    p.add_argument("--test_head", type=int, default=0)
    p.add_argument("--work_threads", type=int, default=8, help="virtual sampler threads (Config.py:65 hard-codes 8)")
    p.add_argument("--seed", type=int, default=0, help="parameter initialisation seed")
    p.add_argument("--sparse_rows", type=int, default=-1, help="1 / 0: force / forbid the touched-rows-only update: TransE int8 records (SGD, or the opt-in non-parity "
                                                             "--optimizer LazyAdam; on N ranks the entity table is sharded by row range), TransH / TransD float records "
                                                             "added to the parameter rows in place (SGD, one process).  Default: automatic for large tables")
    return p.parse_args(argv)


def get_conf(argv):
    '''
    Set the Config class using the program args (distribute_training.py:32-71)
    '''
    con = Config(cpp_lib_path=argv.cpp_lib_path)
    con.set_in_path(argv.input_path)
    con.set_export_files(argv.output_path)
    if argv.mode != 'train':
        con.set_test_link_prediction(True)
    con.set_train_times(argv.train_times)
    con.set_nbatches(argv.n_mini_batches)
    con.set_alpha(argv.alpha)
    con.set_margin(argv.margin)
    con.set_bern(argv.bern_flag)
    if argv.ent_dimension != 0 and argv.rel_dimension != 0:
        con.set_ent_dimension(argv.ent_dimension)
        con.set_rel_dimension(argv.rel_dimension)
        con.hidden_size = argv.ent_dimension
    else:
        con.set_dimension(argv.embedding_dimension)
    con.set_ent_neg_rate(argv.ent_neg_rate)
    con.set_rel_neg_rate(argv.rel_neg_rate)
    con.set_opt_method(argv.optimizer)
    con.set_work_threads(getattr(argv, "work_threads", 8))
    con.seed = getattr(argv, "seed", 0)
    if getattr(argv, "sparse_rows", -1) >= 0:
        con.sparse_rows = bool(argv.sparse_rows)
    con.init()
    name = argv.model.lower()
    con.set_model({"transh": TransH, "transr": TransR, "transd": TransD}.get(name, TransE))
    return con


# ---------------------------------------------------------------------------------------------
# checkpoints: TF Saver naming so that the reference's tooling finds the step
# ---------------------------------------------------------------------------------------------
def get_last_step(output_path):
    '''
    :return: last global step; 0 if there is no checkpoint (distribute_training.py:134-156)
    '''
    last_global_step = 0
    try:
        path = os.path.join(output_path, "checkpoint")
        if os.path.isfile(path):
            with open(path, "r") as f:
                line = f.readline().replace('"', '').split(":")[1].split("/")
                last_global_step = int(line[len(line) - 1].split("-")[1].strip())
    except Exception as e:  # same tolerance as the reference
        print("Error occured during last global step reading:")
        print(e)
    return last_global_step


def _sharded(con):
    return con._dp and getattr(con, "sparse_rows", False) and hasattr(con, "_shard")


def checkpoint_arrays(con):
    """Variables under the reference's names, Adam slots as `<var>/Adam`, `<var>/Adam_1`
    (main_spark.py:74-98), plus the optimiser scalars and the sampler's rng streams.  COLLECTIVE in data-parallel
    runs (owner-kept Adam slots are gathered): every rank calls it, rank 0 writes.  A SHARDED entity table (the
    table-sharded sparse mode) is not gathered: every rank writes its own rows beside the main file (save_checkpoint)."""
    con.sync_optimizer_state()
    if _sharded(con) and con._has_slots:
        from ._lib import KgeError
        raise KgeError("checkpoints of a table-sharded LazyAdam run are not written yet: the moment shards would have to travel with "
                       "the row shards (train with --optimizer SGD on N ranks, or LazyAdam in one process, where checkpoints work)")
    if _sharded(con):
        out = {n: t.detach().cpu().numpy() for n, t in con.trainModel.parameter_lists.items() if n != "ent_embeddings"}
    else:
        out = dict(con.get_parameters())
    if con._has_slots:
        for name, m, v in zip(con.trainModel.table_names, con._adam_m, con._adam_v):
            out[name + "/Adam"] = m.detach().cpu().numpy()
            out[name + "/Adam_1"] = v.detach().cpu().numpy()
        out["beta1_power"] = np.float32(con._beta1_power)
        out["beta2_power"] = np.float32(con._beta2_power)
    out["global_step"] = np.int64(con.global_step)
    # with sampling one step ahead the device streams are one batch further than the training: store the states the
    # NEXT step's batch starts from, so that a resumed run trains on exactly the batches an uninterrupted one would
    out["rng_streams"] = con.get_stream_states(before_prefetch=True)
    return out


def _step_of(path):
    return int(os.path.basename(path).split("model.ckpt-")[1].split(".")[0])


def _atomic_savez(path, **arrays):
    """np.savez under a temporary name, renamed into place: a reader (or a crash) never sees a half-written file."""
    tmp = path + ".tmp%d" % os.getpid()
    with open(tmp, "wb") as f:
        np.savez(f, **arrays)
    os.replace(tmp, path)


def save_checkpoint(con, output_path, max_to_keep=10, write=True):
    """COLLECTIVE in data-parallel runs.  Order of a SHARDED checkpoint (needs an `output_path` every rank can write and, on
    restore, read -- a shared directory): every rank writes its shard file -> barrier -> rank 0 writes the main file and only
    THEN the `checkpoint` pointer, then prunes.  The pointer therefore never names a step whose shard files are incomplete, and
    older complete checkpoints are deleted only after the new one is whole."""
    arrays = checkpoint_arrays(con)
    step = con.global_step
    base = os.path.join(output_path, "model.ckpt-%d" % step)
    if _sharded(con):      # this rank's rows of the entity table: model.ckpt-<step>.shard<g>of<N>.npz
        os.makedirs(output_path, exist_ok=True)
        sh = con._shard
        rows = con.trainModel.parameter_lists["ent_embeddings"][:sh["hi"] - sh["lo"]].detach().cpu().numpy()
        _atomic_savez(base + ".shard%dof%d.npz" % (con.rank, con.world_size), rows=rows, lo=np.int64(sh["lo"]), hi=np.int64(sh["hi"]),
                      ent_total=np.int64(con.entTotal))
        import torch.distributed as dist
        con.comm_fence("pg")
        dist.barrier(group=getattr(con, "_pg", None))      # all shard files of this step exist before anything points at them
    if not write:
        return None
    os.makedirs(output_path, exist_ok=True)
    _atomic_savez(base + ".npz", **{k.replace("/", "__"): v for k, v in arrays.items()})
    tmp = os.path.join(output_path, "checkpoint.tmp%d" % os.getpid())
    with open(tmp, "w") as f:
        f.write('model_checkpoint_path: "%s"\n' % base)
    os.replace(tmp, os.path.join(output_path, "checkpoint"))
    kept = sorted((p for p in glob.glob(os.path.join(output_path, "model.ckpt-*.npz")) if ".shard" not in p), key=_step_of)
    for old in kept[:-max_to_keep]:
        for part in glob.glob(old[:-4] + ".shard*of*.npz"):
            os.remove(part)
        os.remove(old)
    return base + ".npz"


def read_entity_rows(base, lo, hi, dim):
    """Rows [lo, hi) of the entity table of a sharded checkpoint, from whichever shard files hold them (the number of ranks
    may differ from the run that wrote them)."""
    out = np.zeros((hi - lo, dim), np.float32)
    got = 0
    for part in sorted(glob.glob(base + ".shard*of*.npz")):
        z = np.load(part)
        plo, phi = int(z["lo"]), int(z["hi"])
        a, b = max(lo, plo), min(hi, phi)
        if a < b:
            out[a - lo:b - lo] = z["rows"][a - plo:b - plo]
            got += b - a
    if got != hi - lo:
        raise ValueError("sharded checkpoint %s: entity rows [%d, %d) are not all present in its shard files" % (base, lo, hi))
    return out


def grow_table(table, rows, rng, zeros=False):
    """Append rows for new entities (main_spark.py:74-98): xavier-initialised for a parameter table,
    zeros for an Adam slot.  The reference draws the WHOLE [final rows, dim] variable with the xavier
    initializer (main_spark.py:78) and keeps its tail, so the new rows' stddev is sqrt(2.6/(rows+dim))
    with rows = the final row count, not the number of appended rows."""
    table = np.asarray(table, dtype=np.float32)
    extra = rows - table.shape[0]
    if extra <= 0:
        return table
    new = (np.zeros((extra, table.shape[1]), np.float32) if zeros
           else xavier_normal(rng, (extra, table.shape[1]), fan_in=rows))
    return np.concatenate([table, new], axis=0)


def restore_checkpoint(con, path, allow_growth=True, arrays=None):
    """Load a checkpoint into an initialised Config (after set_model_and_session).  If the dataset
    gained entities since the checkpoint was written, entity tables are grown as the reference's
    `update_entities_and_model` does.  `arrays`: the checkpoint's contents when they were read elsewhere
    (rank 0 reads the file and broadcasts it: for replicated tables the other ranks need not see the output directory; a
    SHARDED entity table is read by every rank from the shard files, so that mode needs a shared `output_path`)."""
    import torch
    z = arrays if arrays is not None else {k.replace("__", "/"): v for k, v in np.load(path).items()}
    rng = np.random.default_rng(getattr(con, "seed", 0) + 1)
    shapes = con.trainModel.table_shapes()
    for i, name in enumerate(con.trainModel.table_names):
        rows = shapes[name][0]
        if name not in z:      # a sharded checkpoint: the entity rows live in per-rank shard files beside the main file
            base = path[:-4] if path.endswith(".npz") else path
            if _sharded(con):
                sh = con._shard
                if sh["hi"] > sh["lo"]:
                    part = read_entity_rows(base, sh["lo"], sh["hi"], shapes[name][1])
                    con.trainModel.parameter_lists[name][:sh["hi"] - sh["lo"]].copy_(torch.from_numpy(part))
                    con.tables_changed()
            else:
                con.set_parameters_by_name(name, read_entity_rows(base, 0, rows, shapes[name][1]))
            continue
        tab = z[name]
        if tab.shape[0] != rows:
            if not allow_growth or tab.shape[0] > rows:
                raise ValueError("checkpoint table %s has %d rows, model needs %d" % (name, tab.shape[0], rows))
            tab = grow_table(tab, rows, rng)
        con.set_parameters_by_name(name, tab)
        if con._has_slots and name + "/Adam" in z:
            con._adam_m[i].copy_(torch.from_numpy(grow_table(z[name + "/Adam"], rows, rng, zeros=True)))
            con._adam_v[i].copy_(torch.from_numpy(grow_table(z[name + "/Adam_1"], rows, rng, zeros=True)))
    if con._has_slots and "beta1_power" in z:
        con._beta1_power = np.float32(z["beta1_power"])
        con._beta2_power = np.float32(z["beta2_power"])
    con.global_step = int(z.get("global_step", 0))
    if "rng_streams" in z and len(z["rng_streams"]) == con.workThreads:
        s = np.ascontiguousarray(z["rng_streams"], dtype=np.uint64)
        if torch.cuda.is_available():
            torch.cuda.synchronize()   # a sampler prefetched on the side stream may still be writing the other half of the state buffer
        con.lib.kge_set_stream_states(s.ctypes.data, con.workThreads)
        con._prefetched = None     # a batch drawn ahead belongs to the run that was interrupted
    return con.global_step


# ---------------------------------------------------------------------------------------------
def _init_validation(con, argv):
    """getValidBatch once before the loop (distribute_training.py:262): validation positives and their
    type-constrained negatives, or None when the evaluation files are absent."""
    path = con.in_path if con.in_path.endswith("/") else con.in_path + "/"
    if not all(os.path.exists(path + f) for f in ("valid2id.txt", "test2id.txt", "type_constrain.txt")):
        return None
    import ctypes
    from . import _lib
    L = con.lib
    L.kge_clear_error()
    L.importTestFiles()
    L.importTypeFiles()
    _lib.raise_if_error(L)
    n = L.getValidTotal()
    arrs = [np.zeros(n, np.int64) for _ in range(6)]
    L.getValidBatch.argtypes = [ctypes.c_void_p] * 6
    L.getBestThreshold.argtypes = [ctypes.c_void_p] * 3
    L.getValidBatch(*[a.ctypes.data for a in arrs])
    _lib.raise_if_error(L)
    return arrs


def _validation_accuracy(con, valid):
    ph, pt, pr, nh, nt, nr = valid
    pos = np.ascontiguousarray(con.test_step(ph, pt, pr).reshape(-1), dtype=np.float32)
    neg = np.ascontiguousarray(con.test_step(nh, nt, nr).reshape(-1), dtype=np.float32)
    thresh = np.zeros(con.relTotal, np.float32)
    con.lib.getBestThreshold(thresh.ctypes.data, pos.ctypes.data, neg.ctypes.data)
    correct = (pos <= thresh[pr]).sum() + (neg > thresh[nr]).sum()
    return float(correct) / (2.0 * max(len(pos), 1))


def main_fun(argv):
    """Train or evaluate (distribute_training.py:161-612)."""
    import torch
    distributed = int(os.environ.get("WORLD_SIZE", "1")) > 1
    rank = int(os.environ.get("RANK", "0"))
    if distributed:
        import torch.distributed as dist
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if os.environ.get("KGE_SINGLE_DEVICE") == "1":
            local_rank = 0        # rehearsal of the multi-rank path on a one-GPU box (with KGE_DIST_BACKEND=gloo)
        torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        own_group = not dist.is_initialized()       # a caller that brought its own process group keeps it afterwards
        if own_group:
            dist.init_process_group(os.environ.get("KGE_DIST_BACKEND", "nccl"))   # "nccl" is RCCL on ROCm
    con = get_conf(argv)
    con.device = "cuda:%d" % torch.cuda.current_device()
    con.set_model_and_session(con.model)
    if distributed:
        con.init_distributed()
    last_global_step = get_last_step(argv.output_path) if (argv.output_path and rank == 0) else 0
    arrays = None
    if distributed:   # rank 0 decides: the other ranks need not see the output directory
        box = [last_global_step, None]
        if rank == 0 and last_global_step > 0:
            path = os.path.join(argv.output_path, "model.ckpt-%d.npz" % last_global_step)
            box[1] = {k.replace("__", "/"): v for k, v in np.load(path).items()}
        dist.broadcast_object_list(box, src=0)
        last_global_step, arrays = box
    if last_global_step > 0:
        restore_checkpoint(con, os.path.join(argv.output_path, "model.ckpt-%d.npz" % last_global_step), arrays=arrays)

    if argv.mode != "train":
        if distributed:   # one contiguous slice of the test set per rank, accumulators all-reduced
            metrics = con.link_prediction_distributed(test_head=bool(argv.test_head))
        else:
            out, metrics = con.link_prediction(test_head=bool(argv.test_head))
        if rank == 0:
            print(json.dumps(metrics, indent=1))
            if argv.output_path:
                with open(os.path.join(argv.output_path, "lp_results.json"), "w") as f:
                    json.dump(metrics, f, indent=1)
        return metrics

    valid = _init_validation(con, argv)
    best_acc, wait_steps_acc, best_step_acc = -1.0, 0, last_global_step
    iterations = con.train_times * con.nbatches + last_global_step      # distribute_training.py:205
    patience = argv.early_stop_patience
    stopping_step = argv.early_stop_stopping_step * con.nbatches
    to_reach_step = argv.early_stop_start_step * con.nbatches + last_global_step
    best_loss, wait_steps_loss, best_step = float("inf"), 0, last_global_step
    t0 = time.time()
    g = last_global_step
    while g < iterations:
        if not distributed and con.persistent_preferred():
            # launch-latency-bound step sizes: every step up to the next checkpoint / early-stop check in ONE persistent launch
            # (csrc/persist.hip); the per-step log lines of distribute_training.py:283 are printed from the returned losses
            to_epoch = con.nbatches - (g - last_global_step) % con.nbatches
            chunk = min(iterations - g, to_epoch, max(to_reach_step - g, 1) if g < to_reach_step else to_epoch)
            losses = con.train_steps(chunk)
            for i, l in enumerate(losses):
                gi = g + i + 1
                print('Global step: {} Epoch: {} Batch: {} loss: {}'.format(
                    gi, int((gi - last_global_step) / con.nbatches), int((gi - last_global_step) % con.nbatches), float(l)))
            loss = float(losses[-1])
            g = con.global_step
        else:
            loss = con.train_step()
            g = con.global_step
            if rank == 0:
                print('Global step: {} Epoch: {} Batch: {} loss: {}'.format(
                    g, int((g - last_global_step) / con.nbatches), int((g - last_global_step) % con.nbatches), loss))
        if (g - last_global_step) % con.nbatches == 0 and argv.output_path:
            save_checkpoint(con, argv.output_path, max_to_keep=patience + 5, write=rank == 0)
        if g < iterations and g >= to_reach_step:
            while g >= to_reach_step:
                to_reach_step += stopping_step
            if valid is not None:   # accuracy criterion of distribute_training.py:295-333
                acc = _validation_accuracy(con, valid)
                if argv.debug and rank == 0:
                    print("[ Early Stop Check (Accuracy) ] best %.10f now %.10f" % (best_acc, acc))
                if acc > best_acc:
                    best_acc, wait_steps_acc, best_step_acc = acc, 0, g
                elif wait_steps_acc < patience:
                    wait_steps_acc += 1
                if wait_steps_acc >= patience:
                    if rank == 0:
                        print('Accuracy early stop. Accuracy has not been improved enough in {} times'.format(patience))
                        if argv.output_path:
                            with open(os.path.join(argv.output_path, "stop.txt"), "w") as f:
                                f.write(str(best_step_acc) + "\n")
                    break
            # loss criterion of distribute_training.py:336-360 (every rank sees the same all-reduced loss)
            if loss < best_loss:
                best_loss, wait_steps_loss, best_step = loss, 0, g
            elif wait_steps_loss < patience:
                wait_steps_loss += 1
            if wait_steps_loss >= patience:
                if rank == 0:
                    print('Loss early stop. Losses has not been improved enough in {} times'.format(patience))
                    if argv.output_path:
                        with open(os.path.join(argv.output_path, "stop.txt"), "w") as f:
                            f.write(str(best_step) + "\n")
                break
    if argv.output_path:
        save_checkpoint(con, argv.output_path, max_to_keep=patience + 5, write=rank == 0)
        if rank == 0:
            with open(os.path.join(argv.output_path, "time.txt"), "w") as f:   # main_spark.py:339-344
                f.write(str(time.time() - t0))
    if distributed:
        import torch.distributed as dist
        con.comm_fence("pg")
        dist.barrier()
        if own_group:
            dist.destroy_process_group()
    return con


if __name__ == "__main__":
    main_fun(parse_args())
