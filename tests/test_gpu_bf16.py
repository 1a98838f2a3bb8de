"""The NON-PARITY fast mode `Config.gather_dtype = "bf16"` (BASELINE configs[1] is written with bf16 storage; the reference
itself is fp32, TransE.py:21-22): rows are gathered from bf16 shadows of the fp32 tables.  What must hold:
  * the shadows are exactly the master tables rounded to bf16 (round to nearest even) after every step;
  * master tables, Adam slots and checkpoints stay fp32, and with the mode off nothing changes;
  * the training it does tracks the fp32 parity mode to a STATED tolerance over 20 steps of the bench workload: every step's
    loss within 1 %, and the accumulated update of each table pointing the same way (cosine >= 0.9; Adam's m / (sqrt(v) + eps)
    turns the sign of a near-zero gradient element into a full +-lr step, so single elements are not comparable)."""
import numpy as np
import pytest

from conftest import parity_report

pytestmark = pytest.mark.gpu


def engine(fb_dir, dtype, nbatches=8, n=25, dim=200, opt="Adam"):
    import openkeonspark_amd as pkg
    from openkeonspark_amd import _lib
    _lib.lib().kge_set_option(b"libc_rand_restart", 1)
    con = pkg.Config()
    con.gather_dtype = dtype
    con.set_in_path(fb_dir); con.set_work_threads(8); con.set_bern(1); con.set_dimension(dim); con.set_nbatches(nbatches)
    con.set_ent_neg_rate(n); con.set_alpha(0.001); con.set_opt_method(opt)
    con.init()
    con.set_model_and_session(pkg.TransE)
    return con


def test_bf16_gather_mode_tracks_fp32(fb_dir):
    import torch
    runs = {}
    for dtype in ("fp32", "bf16"):
        con = engine(fb_dir, dtype)
        start = con.get_parameters()
        losses = [con.train_step() for _ in range(20)]
        if dtype == "bf16":
            for master, shadow in zip(con._tables, con._shadow):
                assert shadow.dtype == torch.bfloat16 and torch.equal(master.to(torch.bfloat16), shadow)    # RNE, every row current
            assert all(t.dtype == torch.float32 for t in con._tables + con._adam_m + con._adam_v)
        else:
            assert con._shadow is None
        runs[dtype] = (np.array(losses), con.get_parameters(), start)
        con.lib.kge_transe_set_bf16_shadow(None, None, None, None, None, None)
    l32, p32, start = runs["fp32"]
    l16, p16, _ = runs["bf16"]
    loss_err = float(np.abs(l16 / l32 - 1).max())
    cos = {}
    for k in p32:
        a, b = (p16[k] - start[k]).astype(np.float64).ravel(), (p32[k] - start[k]).astype(np.float64).ravel()
        cos[k] = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b)))
    parity_report("bf16_gather_mode_vs_fp32", steps=20, loss_relerr=loss_err, update_cosine=cos, bound_loss=1e-2, bound_cosine=0.9)
    assert loss_err <= 1e-2 and min(cos.values()) >= 0.9, (loss_err, cos)
    assert l16[-1] < l16[0]


def test_bf16_mode_needs_the_count_path(fb_dir):
    import openkeonspark_amd as pkg
    con = pkg.Config()
    con.gather_dtype = "bf16"
    con.set_in_path(fb_dir); con.set_dimension(50); con.set_nbatches(100)     # width not a multiple of 4
    con.init()
    with pytest.raises(pkg.KgeError):
        con.set_model_and_session(pkg.TransE)
    con2 = pkg.Config()
    con2.gather_dtype = "bf16"
    con2.set_in_path(fb_dir); con2.set_dimension(64); con2.set_nbatches(100)
    con2.init()
    with pytest.raises(pkg.KgeError):
        con2.set_model_and_session(pkg.TransH)
    con2.lib.kge_transe_set_bf16_shadow(None, None, None, None, None, None)
