"""Import the TeMP reference (read-only tree at /root/reference) under the stubs.

TEST INFRASTRUCTURE ONLY -- build-container use (golden generation, oracle
validation).  Nothing here runs on the GPU box: /root/reference does not exist
there.  No reference source is copied; the modules are imported in place.
"""
import argparse
import os
import sys

REF_ROOT = os.environ.get("TEMP_REFERENCE_ROOT", "/root/reference")
_STUBS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_stubs")


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "models"))


def activate():
    """Put stubs + reference on sys.path and chdir to the reference root
    (datasets are opened by relative path, utils/args.py:72)."""
    if not available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    sys.dont_write_bytecode = True
    for p in (REF_ROOT, _STUBS):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.chdir(REF_ROOT)


def make_args(**over):
    """A complete Namespace with every field of utils/args.py:5-65 (+use_cuda)."""
    d = dict(
        dataset_dir="interpolation", dataset="interpolation/icews14", score_function="complex",
        module="GRRGCN", n_gpu=0, distributed_backend="ddp", hidden_size=200, embed_size=200,
        max_nb_epochs=1000, dropout=0.0, rate_lower=0.2, rate_upper=0.8, lambda_1=2, lambda_2=10,
        lambda_3=20, num_layers=1, lr=1e-3, gradient_clip_val=1.0, patience=10, n_bases=100,
        rgcn_layers=2, train_seq_len=8, test_seq_len=8, batch_size=4, seed=123, negative_rate=50,
        num_pos_facts=3000, log_gpu_memory=False, debug=False, rec_only_last_layer=False,
        fast_dev_run=False, use_time_embedding=False, inv_temperature=0.1,
        use_embed_for_non_active=False, edge_dropout=False, random_dropout=False, type1=False,
        post_ensemble=False, post_aggregation=False, learnable_lambda=False, impute=False, EMA=False,
        vote="recency", future=False, filtered=False, all=False, resume=False, model_name=None,
        version=None, config=None, checkpoint_path=None, spatial_checkpoint=None,
        temporal_checkpoint=None, temporal_module="BiGRRGCN", use_cuda=False,
    )
    d.update(over)
    return argparse.Namespace(**d)


def build_graph_dicts(dataset="interpolation/icews14", max_times=None):
    """Per-timestamp (train, val, test) graphs via the reference's own builder
    (utils/dataset.py:151-232,235-251); avoids build_interpolation_graphs which
    pickles into the read-only dataset dir (utils/dataset.py:291-296)."""
    from utils.dataset import (load_quadruples, load_quadruples_interpolation,
                               get_train_val_test_graph_at_t, get_total_number)
    _, total_times = load_quadruples(dataset, 'train.txt', 'valid.txt', 'test.txt')
    time2triples = load_quadruples_interpolation(dataset, 'train.txt', 'valid.txt', 'test.txt', total_times)
    num_e, num_r = get_total_number(dataset, 'stat.txt')
    if max_times is not None:
        total_times = total_times[:max_times]
    tr, va, te = {}, {}, {}
    for tim in total_times:
        g_tr, g_va, g_te = get_train_val_test_graph_at_t(time2triples[tim], num_r)
        tr[tim], va[tim], te[tim] = g_tr, g_va, g_te  # numpy-int keys, as the reference (t.item() is called on them)
    return num_e, num_r, tr, va, te
