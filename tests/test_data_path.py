"""Host-side data path (torch ops, no kernels): CSV loading and normalisation, the thermal-drift model and the
50 %-overlap row windows, against what the reference's own functions gave on a 1,500-row slice of its test split
(fixture g12: train.py load_measured_data / apply_thermal_drift and DPDDataset._prepare_sequences, run by
tools/make_goldens.py)."""
import csv
import importlib.util
import os

import numpy as np
import pytest
import torch

from conftest import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _mod(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "6g-pa-gan-dpd_amd", rel))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _write_csv(path, rows):
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["I", "Q"])
        for a, b in rows:
            w.writerow([repr(float(a)), repr(float(b))])


def test_thermal_drift_matches_reference():
    import dpd_amd  # noqa: F401
    from dpd_amd.data import TEMPERATURE_C, apply_thermal_drift
    g = load_golden("g12_data")
    y = torch.from_numpy(g["y"])
    assert TEMPERATURE_C == {"cold": -20.0, "normal": 25.0, "hot": 70.0}
    for t in (-20, 25, 70):
        got = apply_thermal_drift(y, float(t)).numpy()
        np.testing.assert_allclose(got, g[f"drift_{t}"], rtol=2e-6, atol=1e-7)
    assert np.array_equal(apply_thermal_drift(y, 25.0).numpy(), g["y"])       # no drift at the reference temperature


def test_window_rows_match_reference_sequences():
    import dpd_amd  # noqa: F401
    from dpd_amd.train import window_rows
    g = load_golden("g12_data")
    u = torch.from_numpy(g["u"])
    for L in (256, 100):
        rows = window_rows(u, L + 5)
        assert np.array_equal(rows.numpy(), g[f"win{L}_input"])               # same starts, same count
        assert np.array_equal(rows[:, 5:, :].numpy(), g[f"win{L}_target"])
    with pytest.raises(ValueError):
        window_rows(u[:300], 261)


def test_measured_rows_from_csv(tmp_path):
    """CSV -> normalise by 0.7 / max|u| -> rows; 'all' stacks the cold / normal / hot variants of the PA output."""
    import dpd_amd  # noqa: F401
    from dpd_amd.train import measured_rows, window_rows
    g = load_golden("g12_data")
    _write_csv(tmp_path / "train_input.csv", g["csv_input"])
    _write_csv(tmp_path / "train_output.csv", g["csv_output"])
    x, t = measured_rows(tmp_path, "train", 261, "cpu")
    np.testing.assert_allclose(t.numpy(), g["win256_target"], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(x.numpy(), window_rows(torch.from_numpy(g["y"]), 261).numpy(), rtol=1e-6, atol=1e-8)
    xa, ta = measured_rows(tmp_path, "train", 261, "cpu", temp="all")
    n = x.shape[0]
    assert xa.shape[0] == 3 * n and ta.shape[0] == 3 * n
    np.testing.assert_allclose(xa[n:2 * n].numpy(), x.numpy(), rtol=0, atol=0)
    np.testing.assert_allclose(xa[:n].numpy(), window_rows(torch.from_numpy(g["drift_-20"]), 261).numpy(), rtol=3e-6, atol=1e-7)
    np.testing.assert_allclose(xa[2 * n:].numpy(), window_rows(torch.from_numpy(g["drift_70"]), 261).numpy(), rtol=3e-6, atol=1e-7)
    assert torch.equal(ta[:n], ta[2 * n:])
