"""DUNE training mirror (second half): the loss terms against values produced by the
reference's own `DUNETrain.train_one_epoch(validate=True)` (dune_train.py:302-366; generated in the build
container, see the snippet in tests/golden/make_golden_frontend.py's docstring neighbour
`dune_train_losses.npz`), and -m gpu: a short training run on labels from the HIP labeller."""
import os
import tempfile

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def _golden():
    return np.load(os.path.join(HERE, "golden", "dune_train_losses.npz"))


def test_loss_terms_match_reference_values():
    from neupan_amd.dune_train import DuneTrain, ObsPointNet
    z = _golden()
    net = ObsPointNet(2, 4)
    net.load_state_dict({k[2:]: torch.tensor(z[k]) for k in z.files if k.startswith("w/")})
    net.eval()
    tr = DuneTrain(net, z["G"], z["h"], tempfile.mkdtemp(), device="cpu")
    data = (torch.tensor(z["points"], dtype=torch.float32), torch.tensor(z["mu"], dtype=torch.float32),
            torch.tensor(z["dist"], dtype=torch.float32))
    np.random.seed(11)                                   # one rotation angle per batch, same draw order
    got = np.array(tr._epoch(data, 256, True))
    assert np.allclose(got, z["losses"], rtol=2e-5, atol=0), (got, z["losses"])


def test_state_dict_keys_are_the_reference_ones():
    from neupan_amd.dune_train import ObsPointNet
    z = _golden()
    ref_keys = sorted(k[2:] for k in z.files if k.startswith("w/"))
    assert sorted(ObsPointNet(2, 4).state_dict().keys()) == ref_keys


@pytest.mark.gpu
def test_short_training_run_on_hip_labels():
    from neupan_amd.dune_train import DuneTrain
    from helpers import CONFIGS, make_oracle
    orc = make_oracle(CONFIGS["diff_1k_T10_K10"])
    d = tempfile.mkdtemp()
    torch.manual_seed(0); np.random.seed(0)
    tr = DuneTrain(None, np.asarray(orc.G), np.asarray(orc.h), d)
    full = tr.start(data_size=8000, data_range=[-25, -25, 25, 25], batch_size=256, epoch=40, valid_freq=10, save_freq=40,
                    lr=5e-3, decay_freq=1500)
    assert full.endswith("model_40.pth") and os.path.exists(full)
    assert tr.loss_list[-1] < 0.9 * tr.loss_list[0]     # it learns (slowly: the reference's own log needs ~250 epochs x 313 steps)
    txt = open(os.path.join(d, "results.txt")).read()
    assert "Validate Mu Loss" in txt and "data_size: 8000" in txt
    # the checkpoint is a reference-format state_dict: the planner loads it
    from gpu_helpers import make_gpu_pan
    pan = make_gpu_pan(CONFIGS["corridor_diff_small"], checkpoint=full, iter_num=1)
    assert pan.E == 4


@pytest.mark.gpu
def test_training_workflow_without_checkpoint(tmp_path):
    """The reference's training workflow (example/dune_train/dune_train_diff.yaml: robot + train sections only, no
    dune_checkpoint, train.direct_train: true; dune.py:152-156 builds the planner, then trains): init_from_yaml must
    construct a planner, train_dune() must run and return a checkpoint in the reference's format, planning before that
    must raise (the reference blocks on input()), and a planner built with the new checkpoint must plan."""
    from neupan_amd._lib import NeupanAmdError
    from neupan_amd.planner import neupan
    y = tmp_path / "dune_train_diff.yaml"
    y.write_text("robot:\n  kinematics: 'diff'\n  length: 1.6\n  width: 2.0\n\ntrain:\n  direct_train: true\n  data_size: 4000\n"
                 "  data_range: [-25, -25, 25, 25]\n  batch_size: 256\n  epoch: 6\n  valid_freq: 3\n  save_freq: 6\n  lr: 5e-3\n"
                 f"  lr_decay: 0.5\n  decay_freq: 1500\n  save_dir: '{tmp_path}'\n")
    planner = neupan.init_from_yaml(str(y))
    with pytest.raises(NeupanAmdError):
        planner.pan.forward_batch(np.zeros((1, 3, 11), np.float32), np.zeros((1, 2, 10), np.float32), np.zeros((1, 3, 11), np.float32),
                                  np.zeros((1, 10), np.float32), np.ones((1, 2, 5), np.float32))
    planner.train_dune()
    full = planner.pan.dune_layer.full_model_name
    assert full.endswith("model_6.pth") and os.path.exists(full)
    log = open(os.path.join(os.path.dirname(full), "results.txt")).read()
    assert "Validate Mu Loss" in log and "\\n" not in log and log.count("\n") > 10          # real newlines in the log
    from gpu_helpers import make_gpu_pan
    from helpers import CONFIGS
    pan = make_gpu_pan(CONFIGS["corridor_diff_small"], checkpoint=full, iter_num=1)
    assert pan.E == 4
