"""GPU: the command-line front end (reference flag names) trains / evaluates / writes the reference's artefacts."""
import os
import shutil

import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def test_cli_train_then_test(tmp_path):
    import nbest_amd  # noqa: F401
    from nbest_amd import cli
    root = tmp_path / "data"
    root.mkdir()
    shutil.copy(os.path.join(GOLDEN, "valid_200.txt"), root / "train")
    shutil.copy(os.path.join(GOLDEN, "valid_head.txt"), root / "valid")
    exp = str(tmp_path / "exp")
    common = ["--dataset", "dstc2", "--dataroot", str(root), "--deviceId", "0", "--random_seed", "999", "--dropout", "0.3",
              "--bert_dropout", "0.1", "--optim_choice", "bertadam", "--lr", "3e-5", "--bert_lr", "3e-5", "--warmup_proportion", "0.1",
              "--batchSize", "16", "--max_norm", "5.0", "--max_epoch", "2", "--experiment", exp, "--pre_trained_model", "bert",
              "--coverage", "0.5", "--add_segment_ids", "--add_l2_loss", "--label_space", os.path.join(GOLDEN, "label_space.json"),
              "--vocab", os.path.join(GOLDEN, "text_vocab.json"), "--encoder_layers", "2", "--n_best", "5"]
    assert cli.main(common) == 0
    opt = cli.parse_arguments(common)
    d = cli.exp_dir(opt)
    assert d.endswith("data_dstc2/nl_6__nh_4__dk_64__dv_64__bs_16__dp_0.3_0.1__opt_bertadam_0.1_3e-05_3e-05__mn_5.0__me_2__seed_999__"
                      "score_pp__repr_bin_sa_cls__cls_stc")
    log = open(os.path.join(d, "log.train")).read().split("\n")
    assert sum(l.startswith("[Train]\tEpoch: ") for l in log) == 2 and sum(l.startswith("[Valid]\tEpoch: ") for l in log) == 2
    assert any(l.startswith("BEST RESULT:") for l in log)
    lines = open(os.path.join(d, "valid.iter1")).read().strip("\n").split("\n")
    assert len(lines) == 24 and all(l.count("\t<=>\t") == 2 for l in lines)
    assert lines[0].startswith("[CLS] [SYS] Hello , welcome")
