import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def labels():
    import nbest_amd  # noqa: F401
    from nbest_amd.config import LabelSpace
    return LabelSpace.from_json(os.path.join(GOLDEN, "label_space.json"))


def load_case(name):
    z = np.load(os.path.join(GOLDEN, "case_%s.npz" % name))
    meta = json.loads(str(z["meta"]))
    return meta, z


def case_inputs(meta, labels):
    """Regenerate the inputs of a golden case from its seeds (nothing but outputs is committed)."""
    from nbest_amd import config as ncfg, synth
    mk = {"xlm-roberta": ncfg.xlmr_base, "xlm-roberta-large": ncfg.xlmr_large, "bert": ncfg.bert_base}[meta["family"]]
    cfg = mk(num_hidden_layers=meta["L"], hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    sd = synth.model_state(cfg, labels, seed=meta["seed"])
    if meta.get("outliers"):
        synth.pretrained_like(sd, cfg, seed=meta["seed"], ln_gain=meta.get("ln_gain", 10.0), col_gain=meta.get("col_gain", 20.0))
    batch = synth.nbest_batch(cfg, labels, meta["B"], meta["S"], n_best=meta["n_best"], seed=meta["seed"],
                              ragged=True, trans_len=meta["St"])
    return cfg, sd, batch
