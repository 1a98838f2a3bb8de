"""Model base class: the reference's `Model` protocol (/root/reference/Model.py:5-99) without TensorFlow.

A model is a description of parameter tables plus the id of the fused HIP operator that evaluates
its loss / gradients / predictions.  The attributes the reference's callers use are kept:
``parameter_lists`` (name -> tensor, Model.py:74 and TransE.py:23-24), ``loss``, ``predict``,
``batch_h/t/r/y`` and the (B,1)/(B,N) views of Model.py:62-69 (here as index arithmetic on the flat
device batch: positive b at [b], negative k of positive b at [B*(k+1)+b]).
"""
import numpy as np

from . import _lib


def xavier_normal(rng, shape, fan_in=None):
    """tf.contrib.layers.xavier_initializer(uniform=False) as the reference's get_variable calls use
    it (TransE.py:21-22): truncated normal, stddev sqrt(1.3 * 2 / (fan_in + fan_out)) with
    fan_in = rows, fan_out = cols, values beyond two stddev re-drawn.  `fan_in` overrides the row count
    of `shape` when only a slice of a larger variable is drawn (new-entity rows, main_spark.py:78)."""
    rows, cols = shape
    std = np.sqrt(2.6 / ((rows if fan_in is None else fan_in) + cols))
    a = rng.standard_normal(shape)
    bad = np.abs(a) > 2.0
    while bad.any():
        a[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(a) > 2.0
    return (a * std).astype(np.float32)


LARGE_TABLE_ELEMS = 1 << 28   # beyond 1 GiB of fp32 a table is initialised in HBM, never staged on the host


def xavier_normal_device(shape, device, seed, row_range=None, out_rows=None):
    """Same distribution as xavier_normal, drawn by the device generator in row blocks (a 50M x 512
    table is 102 GB: it exists only in HBM).  row_range=(lo, hi): only those rows are kept, in a tensor of `out_rows`
    rows (a rank's shard of a row-sharded table: the generator still walks the whole table block by block, through one
    512 MB scratch block, so every rank's rows are the rows of the single-process table -- but the table itself never exists)."""
    import torch
    rows, cols = shape
    std = float(np.sqrt(2.6 / (rows + cols)))
    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed))
    block = max(1, (1 << 27) // max(cols, 1))
    if row_range is None:
        out = torch.empty(shape, dtype=torch.float32, device=device)
        for r0 in range(0, rows, block):
            part = out[r0:r0 + block]
            torch.nn.init.trunc_normal_(part, 0.0, std, -2.0 * std, 2.0 * std, generator=gen)   # inverse-CDF draw, in place
        return out
    lo, hi = row_range
    out = torch.zeros((int(out_rows if out_rows is not None else hi - lo), cols), dtype=torch.float32, device=device)
    scratch = torch.empty((min(block, rows), cols), dtype=torch.float32, device=device)
    for r0 in range(0, rows, block):
        r1 = min(r0 + block, rows)
        part = scratch[:r1 - r0]
        torch.nn.init.trunc_normal_(part, 0.0, std, -2.0 * std, 2.0 * std, generator=gen)
        a, b = max(lo, r0), min(hi, r1)
        if b > a:
            out[a - lo:b - lo].copy_(part[a - r0:b - r0])
    del scratch
    return out


class Model(object):
    model_id = None       # _lib.TRANSE ...
    table_names = ()      # engine table order (include/kge_mi355.h)

    def get_config(self):
        return self.config

    # --- batch views (Model.py:10-53).  The reference slices TF placeholders; here they are views of the batch the
    # config currently holds (Config.batch_h/t/r/y as filled by sampling(), layout [B positives | B negs round k ...]):
    # in_batch=True gives the (B, 1) / (B, n) shapes of Model.py:62-69, negative k of positive b at [b, k].
    def _batch(self, name):
        arr = getattr(self, name, None)
        return arr if arr is not None else getattr(self.config, name)

    def _split(self, arr, in_batch, positive):
        B = self.config.batch_size
        if positive:
            return arr[0:B].reshape(1, -1).T if in_batch else arr[0:B]
        rest = arr[B:self.config.batch_seq_size]
        n = self.config.negative_ent + self.config.negative_rel
        return rest.reshape(n, -1).T if in_batch else rest

    def get_positive_instance(self, in_batch=True):
        return [self._split(self._batch(k), in_batch, True) for k in ("batch_h", "batch_t", "batch_r")]

    def get_negative_instance(self, in_batch=True):
        return [self._split(self._batch(k), in_batch, False) for k in ("batch_h", "batch_t", "batch_r")]

    def get_positive_labels(self, in_batch=True):
        return self._split(self._batch("batch_y"), in_batch, True)

    def get_negative_labels(self, in_batch=True):
        return self._split(self._batch("batch_y"), in_batch, False)

    def get_all_instance(self, in_batch=False):
        n = 1 + self.config.negative_ent + self.config.negative_rel
        arrs = [self._batch(k) for k in ("batch_h", "batch_t", "batch_r")]
        return [a.reshape(n, -1).T for a in arrs] if in_batch else arrs

    def get_all_labels(self, in_batch=False):
        n = 1 + self.config.negative_ent + self.config.negative_rel
        y = self._batch("batch_y")
        return y.reshape(n, -1).T if in_batch else y

    def get_predict_instance(self):
        return [self.predict_h, self.predict_t, self.predict_r]

    # --- Model.py:55-74 -------------------------------------------------------------------------
    def input_def(self):
        config = self.config
        # flat batch of batch_seq_size ids; filled by Config.sampling()/train_step
        self.batch_h = None
        self.batch_t = None
        self.batch_r = None
        self.batch_y = None
        self.predict_h = None
        self.predict_t = None
        self.predict_r = None
        self.parameter_lists = {}
        self.batch_seq_size = getattr(config, "batch_seq_size", None)

    def table_shapes(self):
        raise NotImplementedError

    def embedding_def(self):
        """Allocate and initialise the parameter tables on the device (names = checkpoint contract)."""
        import torch
        config = self.config
        rng = np.random.default_rng(getattr(config, "seed", 0))
        device = getattr(config, "device", "cuda")
        self.parameter_lists = {}
        plan = getattr(config, "_shard_plan", None)     # row-sharded entity table (Config._plan_entity_shard): this rank's rows only
        for name in self.table_names:
            shape = self.table_shapes()[name]
            shard = plan if (plan is not None and name == "ent_embeddings") else None
            if int(np.prod(shape)) > LARGE_TABLE_ELEMS:
                if shard is not None:
                    self.parameter_lists[name] = xavier_normal_device(shape, device, getattr(config, "seed", 0),
                                                                      row_range=(shard["lo"], shard["hi"]), out_rows=shard["chunk"])
                else:
                    self.parameter_lists[name] = xavier_normal_device(shape, device, getattr(config, "seed", 0))
            else:
                full = xavier_normal(rng, shape)
                if shard is not None:
                    part = np.zeros((shard["chunk"], shape[1]), np.float32)
                    part[:shard["hi"] - shard["lo"]] = full[shard["lo"]:shard["hi"]]
                    full = part
                self.parameter_lists[name] = torch.from_numpy(full).to(device)
        config._ent_is_shard = plan is not None
        for name, t in self.parameter_lists.items():
            setattr(self, name, t)

    def loss_def(self):
        self.loss = None   # set by every train step (device scalar)

    def predict_def(self):
        self.predict = None  # set by Config.test_step

    def descriptor(self):
        c = self.config
        ent_dim, rel_dim = self.dims()
        return _lib.ModelDesc(self.model_id, int(c.negative_rel), int(c.entTotal), int(c.relTotal),
                              int(ent_dim), int(rel_dim), float(c.margin), 0)

    def dims(self):
        c = self.config
        return c.hidden_size, c.hidden_size

    def __init__(self, config, define=False):
        self.config = config
        self.parameter_lists = {}
        if define:
            self.input_def()
            self.embedding_def()
            self.loss_def()
            self.predict_def()
