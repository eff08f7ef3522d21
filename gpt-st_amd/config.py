"""Flag/config system for ``-mode pretrain`` (mirrors reference lib/Params_pretrain.py:6-77).

Two-stage parse like the reference: stage 1 reads ``-dataset/-mode/-device/-model``,
stage 2 loads ``conf/GPTST_pretrain/<dataset>.conf`` and registers every key as a
single-dash flag whose default is the INI value.  Unlike the reference the conf path is
resolved relative to this package (not the CWD) and no predictor conf is read
(reference Run.py:36 reads one and never uses it in pretrain mode).
"""
import argparse
import configparser
import os
from types import SimpleNamespace

CONF_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "conf", "GPTST_pretrain")

# dataset -> (interval minutes, week_day) as set by reference lib/load_dataset.py:50-53,61-64,73-77,85-89
DATASET_TIME = {"PEMS08": (5, 7), "METR_LA": (5, 7), "NYC_BIKE": (30, 7), "NYC_TAXI": (30, 7)}

_EVAL = lambda s: eval(s) if isinstance(s, str) else s  # noqa: E731  (reference uses type=eval for bools/None)

# (section, key, type) in the reference's registration order (Params_pretrain.py:26-74)
_KEYS = [
    ("data", "val_ratio", float), ("data", "test_ratio", float), ("data", "lag", int), ("data", "horizon", int),
    ("data", "num_nodes", int), ("data", "tod", _EVAL), ("data", "normalizer", str), ("data", "column_wise", _EVAL),
    ("data", "default_graph", _EVAL),
    ("model", "input_base_dim", int), ("model", "input_extra_dim", int), ("model", "output_dim", int),
    ("model", "embed_dim", int), ("model", "embed_dim_spa", int), ("model", "hidden_dim", int), ("model", "HS", int),
    ("model", "HT", int), ("model", "HT_Tem", int), ("model", "num_route", int), ("model", "mask_ratio", float),
    ("model", "ada_mask_ratio", float), ("model", "ada_type", str),
    ("train", "loss_func", str), ("train", "seed", int), ("train", "batch_size", int), ("train", "epochs", int),
    ("train", "lr_init", float), ("train", "lr_decay", _EVAL), ("train", "lr_decay_rate", float),
    ("train", "lr_decay_step", str), ("train", "early_stop", _EVAL), ("train", "early_stop_patience", int),
    ("train", "change_epoch", int), ("train", "up_epoch", str), ("train", "grad_norm", _EVAL),
    ("train", "max_grad_norm", int), ("train", "debug", _EVAL), ("train", "real_value", _EVAL),
    ("train", "seed_mode", _EVAL), ("train", "xavier", _EVAL), ("train", "load_pretrain_path", str),
    ("train", "save_pretrain_path", str),
    ("test", "mae_thresh", _EVAL), ("test", "mape_thresh", float),
    ("log", "log_step", int), ("log", "plot", _EVAL),
]


def _read_conf(dataset):
    path = os.path.join(CONF_DIR, "%s.conf" % dataset)
    if not os.path.isfile(path):
        raise FileNotFoundError("no pretrain conf for dataset %r (looked in %s)" % (dataset, CONF_DIR))
    cp = configparser.ConfigParser()
    cp.optionxform = str  # keep HS/HT/HT_Tem case
    cp.read(path)
    return cp


def parse_args(device, argv=None):
    """Same flag surface as reference ``parse_args(device)`` for the pretrain set."""
    ap = argparse.ArgumentParser(prefix_chars="-", description="pretrain_arguments")
    ap.add_argument("-dataset", default="METR_LA", type=str, required=True)
    ap.add_argument("-mode", default="ori", type=str, required=True)
    ap.add_argument("-device", default=device, type=str)
    ap.add_argument("-model", default="TGCN", type=str)
    ap.add_argument("-cuda", default=True, type=bool)
    first, _ = ap.parse_known_args(argv)
    cp = _read_conf(first.dataset)
    for sec, key, typ in _KEYS:
        ap.add_argument("-" + key, default=typ(cp[sec][key]), type=typ)
    ap.add_argument("-log_dir", default="./", type=str)
    ap.add_argument("-steps_per_replay", default=STEPS_PER_REPLAY, type=int)      # not a reference option: see STEPS_PER_REPLAY
    args, _ = ap.parse_known_args(argv)
    args.interval, args.week_day = DATASET_TIME.get(args.dataset, (5, 7))
    return args


# Optimisation steps enqueued per hipGraph replay by the trainer and bench.py (PretrainStep.step_group): the device idles ~19 us between
# two graph replays, one replay per 4 steps removes three quarters of that.  1 = one replay per step.  Results do not depend on it.
STEPS_PER_REPLAY = int(os.environ.get("GPTST_STEPS_PER_REPLAY", "4"))


def make_args(dataset="PEMS08", mode="pretrain", device="cpu", **overrides):
    """Programmatic equivalent of ``parse_args`` (used by tests, bench and golden scripts)."""
    cp = _read_conf(dataset)
    ns = SimpleNamespace(dataset=dataset, mode=mode, device=device, model="TGCN", cuda=True, log_dir="./")
    for sec, key, typ in _KEYS:
        setattr(ns, key, typ(cp[sec][key]))
    ns.interval, ns.week_day = DATASET_TIME.get(dataset, (5, 7))
    ns.scaler_zeros = 0.0
    ns.steps_per_replay = STEPS_PER_REPLAY
    for k, v in overrides.items():
        setattr(ns, k, v)
    return ns


# ---- downstream predictors (-mode eval): reference lib/Params_predictor.py:4-25 + model/STGCN/args.py:52-76 -------------------------
PRED_CONF = os.path.join(os.path.dirname(CONF_DIR), "GPTST_pretrain", "params_predictors.conf")
_PRED_TRAIN = [("batch_size", int), ("epochs", int), ("lr_init", float), ("lr_decay", _EVAL), ("lr_decay_rate", float), ("lr_decay_step", str),
               ("early_stop", _EVAL), ("early_stop_patience", int), ("grad_norm", _EVAL), ("max_grad_norm", int), ("debug", _EVAL),
               ("real_value", _EVAL)]
_STGCN_KEYS = [("data", "num_nodes", int), ("data", "input_window", int), ("data", "output_window", int), ("model", "Ks", int),
               ("model", "Kt", int), ("model", "blocks1", _EVAL), ("model", "drop_prob", int), ("model", "outputl_ks", int),
               ("train", "seed", int), ("train", "seed_mode", _EVAL), ("train", "xavier", _EVAL), ("train", "loss_func", str)]


def predictor_args(dataset, model="STGCN", argv=None):
    """The predictor's own argument set (double-dash flags as in the reference): the shared training schedule of
    params_predictors.conf, then conf/<model>/<dataset>.conf.  Only STGCN is built."""
    if model != "STGCN":
        raise ValueError("gpt-st_amd builds the STGCN predictor only, got %r" % (model,))
    cp = configparser.ConfigParser()
    cp.optionxform = str
    cp.read(PRED_CONF)
    ap = argparse.ArgumentParser(prefix_chars="--", description="predictor_based_arguments")
    for key, typ in _PRED_TRAIN:
        ap.add_argument("--" + key, default=typ(cp["train"][key]), type=typ)
    path = os.path.join(os.path.dirname(CONF_DIR), model, "%s.conf" % dataset)
    if not os.path.isfile(path):
        raise FileNotFoundError("no %s conf for dataset %r (%s)" % (model, dataset, path))
    cm = configparser.ConfigParser()
    cm.optionxform = str
    cm.read(path)
    for sec, key, typ in _STGCN_KEYS:
        ap.add_argument("--" + key, default=typ(cm[sec][key]), type=typ)
    pargs, _ = ap.parse_known_args([] if argv is None else argv)
    return pargs


def apply_predictor_overrides(args, pargs):
    """reference Run.py:36-43: outside pretrain mode every attribute the predictor's argument set also has replaces the pretrain
    conf's value (batch_size 64, epochs 100, lr_decay_step 25,50,75, early_stop_patience 25, debug False, xavier False, seed, ...)."""
    for attr in list(vars(args)):
        if hasattr(pargs, attr):
            setattr(args, attr, getattr(pargs, attr))
    return args
