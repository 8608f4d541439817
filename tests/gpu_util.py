"""Helpers shared by the `-m gpu` parity tests (HIP path vs the CPU oracle)."""
import torch

from afm import synth


def dev():
    return torch.device("cuda:0")


def to_dev(sd):
    return {k: v.to(dev()) for k, v in sd.items()}


def report(name, got, want, tol):
    got, want = got.detach().float().cpu(), want.detach().float().cpu()
    assert got.shape == want.shape, (name, got.shape, want.shape)
    assert torch.isfinite(got).all(), f"{name}: non-finite values"
    err = (got - want).abs().max().item()
    ref = want.abs().max().item()
    print(f"[parity] {name}: max|diff|={err:.3e} (max|ref|={ref:.3e}, tol={tol:.1e})")
    assert err <= tol, f"{name}: max abs err {err:.3e} > {tol:.1e}"
    return err


def load_named_weights(module, seed=synth.WEIGHT_SEED):
    """Fill a product nn.Module with the name-keyed deterministic weights (same values the oracle's
    shape tables produce for the same keys)."""
    synth.fill_module_(module, seed)
    return module
