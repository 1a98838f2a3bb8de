"""Training driver: CLI flags, checkpoint naming (the reference's get_last_step must find the step),
new-entity growth, and -- on the GPU -- save / restore / resume equivalence and early stop."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from openkeonspark_amd import distribute_training as dt


def test_flags_match_the_reference_cli():
    # main_spark.py:299-324
    a = dt.parse_args(["--input_path", "x/", "--output_path", "y/", "--model", "TransH", "--optimizer", "Adam",
                       "--ent_neg_rate", "25", "--n_mini_batches", "4", "--bern_flag", "1", "--test_head", "1"])
    assert (a.train_times, a.alpha, a.margin, a.embedding_dimension, a.rel_neg_rate, a.early_stop_patience) == \
        (100, 0.00001, 1.0, 64, 0, 5)
    assert a.model == "TransH" and a.optimizer == "Adam" and a.ent_neg_rate == 25 and a.mode == "train"


def test_get_last_step_parses_tf_style_checkpoint_file(tmp_path):
    assert dt.get_last_step(str(tmp_path)) == 0
    (tmp_path / "checkpoint").write_text('model_checkpoint_path: "/some/dir/model.ckpt-4200"\nall_model_checkpoint_paths: "x"\n')
    assert dt.get_last_step(str(tmp_path)) == 4200   # distribute_training.py:134-156


def test_grow_table_appends_xavier_or_zero_rows():
    rng = np.random.default_rng(0)
    t = np.ones((5, 8), np.float32)
    g = dt.grow_table(t, 9, rng)
    assert g.shape == (9, 8) and np.array_equal(g[:5], t) and np.abs(g[5:]).max() > 0
    # main_spark.py:78 draws the whole [final rows, dim] variable: stddev sqrt(2.6 / (9 + 8)), truncated at two stddev
    assert np.abs(g[5:]).max() <= 2 * np.sqrt(2.6 / (9 + 8)) + 1e-6
    big = dt.grow_table(np.ones((14541, 64), np.float32), 14551, np.random.default_rng(1))
    assert abs(big[14541:].std() / (0.88 * np.sqrt(2.6 / (14551 + 64))) - 1) < 0.15   # 0.88 = std of a 2-sigma truncated normal
    z = dt.grow_table(t, 9, rng, zeros=True)
    assert not z[5:].any()
    assert dt.grow_table(t, 5, rng) is not None and dt.grow_table(t, 3, rng).shape == (5, 8)


@pytest.mark.gpu
@pytest.mark.parametrize("opt", ["SGD", "Adam", "LazyAdam"])   # LazyAdam: the opt-in non-parity touched-rows Adam (its moments and powers travel too)
def test_checkpoint_resume_is_bit_identical(tmp_path, opt, monkeypatch):
    monkeypatch.setenv("KGE_COUNTS_MIN_RECORDS", "0")   # the exact count pipeline: reproducible bit for bit at any step size
    out = str(tmp_path / "run")
    base = ["--input_path", os.path.join(GOLDEN, "kg_small"), "--output_path", out, "--embedding_dimension", "32",
            "--n_mini_batches", "5", "--ent_neg_rate", "3", "--alpha", "0.01", "--optimizer", opt, "--bern_flag", "1"]
    from openkeonspark_amd import _lib
    fresh = lambda: _lib.lib().kge_set_option(b"libc_rand_restart", 1)   # each run below stands for a new process
    # 4 epochs in one go
    fresh()
    full = dt.main_fun(dt.parse_args(base + ["--train_times", "4", "--output_path", str(tmp_path / "full")]))
    want = full.get_parameters()
    # 2 epochs, then a NEW process-like session that resumes from the checkpoint for 2 more
    fresh()
    a = dt.main_fun(dt.parse_args(base + ["--train_times", "2"]))
    assert dt.get_last_step(out) == 10 and a.global_step == 10
    fresh()
    b = dt.main_fun(dt.parse_args(base + ["--train_times", "2"]))
    assert b.global_step == 20 and dt.get_last_step(out) == 20
    got = b.get_parameters()
    for k in want:
        assert np.array_equal(got[k], want[k]), k
    assert os.path.exists(os.path.join(out, "time.txt"))


@pytest.mark.gpu
def test_restore_grows_new_entities(tmp_path):
    import openkeonspark_amd as pkg
    def make(E):
        con = pkg.Config()
        con.set_dimension(16); con.set_opt_method("Adam")
        hh = np.arange(60) % 50
        con.init_from_arrays(E, 4, hh, (hh + 1) % 50, hh % 4)
        con.set_model_and_session(pkg.TransE)
        return con
    old = make(50)
    old.train_step()
    path = dt.save_checkpoint(old, str(tmp_path))
    new = make(57)                                       # 7 entities arrived (main_spark.py:29-114)
    dt.restore_checkpoint(new, path)
    p_old, p_new = old.get_parameters(), new.get_parameters()
    assert np.array_equal(p_new["ent_embeddings"][:50], p_old["ent_embeddings"])
    assert np.abs(p_new["ent_embeddings"][50:]).max() > 0
    assert np.array_equal(p_new["rel_embeddings"], p_old["rel_embeddings"])
    assert not new._adam_m[0][50:].any().item() and new.global_step == 1
    new.train_step()                                      # and training continues on the grown tables


@pytest.mark.gpu
def test_loss_early_stop_writes_stop_file(tmp_path):
    out = str(tmp_path / "es")
    args = dt.parse_args(["--input_path", os.path.join(GOLDEN, "kg_tiny"), "--output_path", out, "--embedding_dimension", "8",
                          "--n_mini_batches", "2", "--alpha", "0.0", "--train_times", "50", "--early_stop_patience", "2"])
    con = dt.main_fun(args)                               # lr = 0: the loss never improves after the first check
    assert os.path.exists(os.path.join(out, "stop.txt")) and con.global_step < 50 * con.nbatches


@pytest.mark.gpu
def test_accuracy_early_stop_uses_triple_classification(tmp_path, capsys):
    out = str(tmp_path / "acc")
    args = dt.parse_args(["--input_path", os.path.join(GOLDEN, "kg_small"), "--output_path", out, "--embedding_dimension", "16",
                          "--n_mini_batches", "2", "--alpha", "0.0", "--train_times", "40", "--early_stop_patience", "2",
                          "--debug", "1"])
    con = dt.main_fun(args)   # lr = 0: neither accuracy nor loss can improve -> stops after `patience` checks
    text = capsys.readouterr().out
    assert "Early Stop Check (Accuracy)" in text and "early stop" in text
    assert os.path.exists(os.path.join(out, "stop.txt")) and con.global_step < 40 * con.nbatches
