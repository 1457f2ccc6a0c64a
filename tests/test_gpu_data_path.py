"""The data path on the device: thermal drift, 50 %-overlap rows and CSV -> rows with device="cuda" against what the
reference's own functions gave (fixture g12: train.py load_measured_data / apply_thermal_drift and
DPDDataset._prepare_sequences, run by tools/make_goldens.py), and one run of the training harness from a
`--data-dir` made of that CSV slice."""
import csv

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _write_csv(path, rows):
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["I", "Q"])
        for a, b in rows:
            w.writerow([repr(float(a)), repr(float(b))])


def test_thermal_drift_on_device_matches_reference():
    from dpd_amd.data import apply_thermal_drift
    g = load_golden("g12_data")
    y = torch.from_numpy(g["y"]).to(DEV)
    for t in (-20, 25, 70):
        got = apply_thermal_drift(y, float(t))
        assert got.is_cuda
        np.testing.assert_allclose(got.cpu().numpy(), g[f"drift_{t}"], rtol=2e-6, atol=1e-7)


def test_window_rows_on_device_match_reference_sequences():
    from dpd_amd.train import window_rows
    g = load_golden("g12_data")
    u = torch.from_numpy(g["u"]).to(DEV)
    for L in (256, 100):
        rows = window_rows(u, L + 5)
        assert rows.is_cuda and rows.is_contiguous()
        assert np.array_equal(rows.cpu().numpy(), g[f"win{L}_input"])
        assert np.array_equal(rows[:, 5:, :].cpu().numpy(), g[f"win{L}_target"])
    with pytest.raises(ValueError):
        window_rows(u[:300], 261)


def test_measured_rows_on_device(tmp_path):
    from dpd_amd.train import measured_rows, window_rows
    g = load_golden("g12_data")
    _write_csv(tmp_path / "train_input.csv", g["csv_input"])
    _write_csv(tmp_path / "train_output.csv", g["csv_output"])
    x, t = measured_rows(tmp_path, "train", 261, DEV)
    assert x.is_cuda and t.is_cuda
    np.testing.assert_allclose(t.cpu().numpy(), g["win256_target"], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(x.cpu().numpy(), window_rows(torch.from_numpy(g["y"]), 261).numpy(), rtol=1e-6, atol=1e-8)
    xa, ta = measured_rows(tmp_path, "train", 261, DEV, temp="all")
    n = x.shape[0]
    assert xa.shape[0] == 3 * n and ta.shape[0] == 3 * n
    assert torch.equal(xa[n:2 * n], x)
    np.testing.assert_allclose(xa[:n].cpu().numpy(), window_rows(torch.from_numpy(g["drift_-20"]), 261).numpy(), rtol=3e-6, atol=1e-7)
    np.testing.assert_allclose(xa[2 * n:].cpu().numpy(), window_rows(torch.from_numpy(g["drift_70"]), 261).numpy(), rtol=3e-6, atol=1e-7)


def test_training_harness_from_data_dir(tmp_path):
    """train.main(--data-dir ...): the CSV slice becomes rows on the device, one supervised and one CWGAN-GP epoch run on
    the HIP kernels (QAT on), validation reads the val split, the checkpoint files of the reference are written."""
    from dpd_amd.train import main
    g = load_golden("g12_data")
    d = tmp_path / "data"
    d.mkdir()
    for split in ("train", "val"):
        _write_csv(d / f"{split}_input.csv", g["csv_input"])
        _write_csv(d / f"{split}_output.csv", g["csv_output"])
    out = tmp_path / "ck"
    m = main(["--data-dir", str(d), "--epochs", "2", "--stage1-epochs", "1", "--seq-len", "261", "--batch-size", "8",
              "--temp", "all", "--qat", "--augment", "0.1", "--output-dir", str(out)])
    assert m["stage"] == 2
    for k in ("train_loss", "val_evm_db", "val_nmse_db", "val_l1"):
        assert np.isfinite(m[k]), (k, m[k])
    assert (out / "latest.pth").exists() and (out / "best.pth").exists()
    ck = torch.load(out / "latest.pth", map_location="cpu", weights_only=True)
    assert ck["epoch"] == 1 and "generator_state_dict" in ck and "discriminator_state_dict" in ck
