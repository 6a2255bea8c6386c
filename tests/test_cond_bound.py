"""The per-site conditioned float32 bound (tests/cond_bound.py) against the REFERENCE's own float32 outputs
(tests/golden/ref_fp32.npz, written by make_golden_fp32.py from the imported reference) and against the oracle run in
float32: both must sit inside C_SITE x bound at every site of every RQ-spline golden, forward and inverse.  The GPU
tests (test_gpu_parity.py) hold the HIP kernels to the same bound with the same constant, so "the kernel is no worse than
the reference's float32, site by site" is a tested statement, not a tolerance."""
import numpy as np
import pytest
import torch

from oracle import nf_oracle as O
from test_oracle_golden import ATOM_OPTS, atom_cases
import cond_bound as CB

RQS_CASES = [t for t in atom_cases() if t.split("/")[0].startswith("rqs") or t.startswith("multirqs")]


def _opts(z, tag, dtype=torch.float64):
    kind = tag.split("/")[0]
    opts = dict(ATOM_OPTS[kind])
    if kind == "rqs_fixedx":
        opts["knots_x"] = torch.from_numpy(z[f"{tag}/knots_x"]).to(dtype)
    return opts


def _oracle32_sites(z, tag, inverse):
    """per-site (value, log-derivative) of the oracle's float32 evaluation: (B, S, n) arrays."""
    with torch.device("cpu"):
        return _oracle32_sites_cpu(z, tag, inverse)


def _oracle32_sites_cpu(z, tag, inverse):
    kind = tag.split("/")[0]
    opts = _opts(z, tag, torch.float32)
    b = CB.case_bounds(z, tag, _opts(z, tag), inverse)       # only for the index bookkeeping
    am, Cs = b["am"], b["C"]
    inp = np.asarray(z[f"{tag}/y" if inverse else f"{tag}/x_active"])
    B = inp.shape[0]
    S = 2 if kind == "multirqs" else 1
    inp = torch.from_numpy(inp.reshape(B, S, -1)[:, :, am]).float()
    out = torch.from_numpy(np.asarray(z[f"{tag}/out"]).reshape(B, S, Cs, -1)[:, :, :, am]).float()
    vals, logs = [], []
    for s in range(S):
        if kind == "multirqs":
            o = dict(xlim=opts["xlims"][s], ylim=opts["ylims"][s], extrap=opts["extraps"][s])
        else:
            o = dict(opts)
        kx, ky, kd = O.knots_from_logits(out[:, s], o["xlim"], o["ylim"], o.get("knots_x"))
        kx, ky, kd = (O._bcast_like(k, out[:, s]) for k in (kx, ky, kd))
        full = (B, kd.shape[1], out.shape[-1])
        kx, ky, kd = (k.expand(full) for k in (kx, ky, kd))
        kx, ky, kd = O.augment_knots(kx, ky, kd, axis=1, **(o["extrap"] or {}))
        f = O.rqs_invert if inverse else O.rqs_evaluate
        v, g = f(kx, ky, kd, inp[:, s].unsqueeze(1), axis=1)
        vals.append(v.squeeze(1).double().numpy())
        logs.append(torch.log(g).squeeze(1).double().numpy())
    return np.stack(vals, 1), np.stack(logs, 1)


@pytest.mark.parametrize("tag", RQS_CASES)
def test_reference_float32_outputs_sit_inside_the_conditioned_bound(golden, parity_report, tag):
    z, r = golden("atoms"), golden("ref_fp32")
    b = CB.case_bounds(z, tag, _opts(z, tag))
    B, S, n = b["val"].shape
    am = b["am"]
    # the bound's own float64 evaluation is the golden (the segment formula restated once more)
    y64 = np.asarray(z[f"{tag}/y"]).reshape(B, S, -1)[:, :, am]
    assert np.abs(b["val"] - y64).max() <= 1e-12 * max(1.0, np.abs(y64).max())
    lj64 = np.asarray(z[f"{tag}/logJ"]) - np.asarray(z[f"{tag}/log0"])
    assert np.abs(b["logd"].sum(axis=(1, 2)) - lj64).max() <= 1e-11 * max(1.0, np.abs(lj64).max())
    # the reference's float32 run: y per site, log|J| per sample
    yr = r[f"atoms/{tag}/y"].astype(np.float64).reshape(B, S, -1)[:, :, am]
    ry = (np.abs(yr - b["val"]) / b["b_val"]).max()
    ej = np.abs(r[f"atoms/{tag}/logJ"].astype(np.float64) - np.asarray(z[f"{tag}/logJ"]))
    bj = b["b_logd"].sum(axis=(1, 2)) + CB.EPS32 * np.abs(np.asarray(z[f"{tag}/logJ"]))
    rj = (ej / bj).max()
    parity_report(tag, "ref-fp32 y/site", ry, CB.C_SITE, "largest err / site bound")
    iw = int(np.argmax(ej / bj))
    parity_report(tag, "ref-fp32 logJ/sample", ej[iw], CB.C_SITE * bj[iw], "the sample with the largest err / bound")
    assert ry <= CB.C_SITE and rj <= CB.C_SITE, (tag, ry, rj)
    # the oracle in float32, site by site, both quantities
    v32, l32 = _oracle32_sites(z, tag, False)
    rv, rl = (np.abs(v32 - b["val"]) / b["b_val"]).max(), (np.abs(l32 - b["logd"]) / b["b_logd"]).max()
    assert rv <= CB.C_SITE and rl <= CB.C_SITE, (tag, rv, rl)


@pytest.mark.parametrize("tag", RQS_CASES)
def test_float32_inverse_sits_inside_the_conditioned_bound(golden, parity_report, tag):
    """The inverse: the reference's own float32 inverse is no yardstick (its root cancels), so the
    float32 evaluation here is the oracle's stable root.  The bound's float64 inverse must reproduce the golden x."""
    z = golden("atoms")
    b = CB.case_bounds(z, tag, _opts(z, tag), inverse=True)
    B, S, n = b["val"].shape
    x64 = np.asarray(z[f"{tag}/x_active"]).reshape(B, S, -1)[:, :, b["am"]]
    assert np.abs(b["val"] - x64).max() <= 1e-8
    v32, l32 = _oracle32_sites(z, tag, True)
    rv, rl = (np.abs(v32 - b["val"]) / b["b_val"]).max(), (np.abs(l32 - b["logd"]) / b["b_logd"]).max()
    parity_report(tag, "oracle-fp32 xhat/site", rv, CB.C_SITE, "largest err / site bound")
    assert rv <= CB.C_SITE and rl <= CB.C_SITE, (tag, rv, rl)
