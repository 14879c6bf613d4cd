"""ORACLE -- golden vectors of the episode feed (run in the BUILD CONTAINER only; needs /root/reference).

    python -m oracle.gen_golden_data            # writes tests/golden/data_feed.npz

Imports the reference's OWN ``models/transform.py`` and ``models/dataset.py`` (package ``__init__``s bypassed as in
``gen_golden.py``; stand-ins only for the absent ``gdown`` / ``lightning``), runs every transform class, the observation
shape normaliser, the 80/20 split and ``EpisodeDataset`` + ``StackDataset`` + ``DataLoader`` (shuffle off, workers 0, noise
seeded) on seeded tensors written to a temporary directory, checks ``oracle/ref_data.py`` against them, and freezes inputs and
outputs.  Fixtures hold data only.
"""

from __future__ import annotations

import importlib
import sys
import tempfile
import types
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
REF_SRC = Path("/root/reference/src")
sys.path.insert(0, str(ROOT))

from oracle import ref_data  # noqa: E402


def main() -> None:
    sys.path.insert(0, str(ROOT / "oracle" / "standins"))
    for pkg in ("multimodal_rssm", "multimodal_rssm.models"):
        mod = types.ModuleType(pkg)
        mod.__path__ = [str(REF_SRC / pkg.replace(".", "/"))]
        sys.modules[pkg] = mod
    tr = importlib.import_module("multimodal_rssm.models.transform")
    ds = importlib.import_module("multimodal_rssm.models.dataset")
    from torch.utils.data import DataLoader, StackDataset

    g = torch.Generator().manual_seed(2024)
    out: dict[str, np.ndarray] = {}
    vis = torch.randint(0, 256, (7, 3, 8, 8), generator=g).float()
    aud = torch.rand(7, 1, 8, 4, generator=g) * 80.1 - 80.0
    act = torch.randn(7, 6, generator=g)
    out["in/vision"], out["in/audio"], out["in/action"] = vis.numpy(), aud.numpy(), act.numpy()
    out["out/normalize_vision"] = tr.NormalizeVisionImage()(vis).numpy()
    out["out/normalize_audio_default"] = tr.NormalizeAudioMelSpectrogram()(aud).numpy()
    out["out/normalize_audio_yaml"] = tr.NormalizeAudioMelSpectrogram(min_value=-80.0, max_value=0.0)(aud).numpy()
    out["out/take_first_5"] = tr.TakeFirstN(5)(act).numpy()
    out["out/remove_dim"] = tr.RemoveDim(1, [0, 4])(act).numpy()
    torch.manual_seed(99)
    out["out/gaussian_noise"] = tr.GaussianNoise(0.1)(act).numpy()
    out["out/norm_shape_5d"] = ds.BaseEpisodeDataModule._normalize_observation_shape(torch.arange(2 * 3 * 4 * 5 * 2.0).reshape(2, 3, 4, 5, 2)).numpy()  # noqa: SLF001
    out["out/norm_shape_4d"] = ds.BaseEpisodeDataModule._normalize_observation_shape(torch.arange(2 * 3 * 4 * 5.0).reshape(2, 3, 4, 5)).numpy()  # noqa: SLF001
    for n in (1, 4, 5, 10, 11):
        a, b = ds.split_path_list([Path(f"p{i}") for i in range(n)], 0.8)
        out[f"out/split_{n}"] = np.array([len(a), len(b)])
        assert ref_data.split_path_list(list(range(n)), 0.8) == (list(range(len(a))), list(range(len(a), n)))

    # ---- the 6-tuple batches of EpisodeDataset + StackDataset + DataLoader, noise seeded, on 5 episodes of 9 steps
    n_ep, t_full, t_take, bs = 5, 9, 6, 2
    eps = {"act": torch.randn(n_ep, t_full, 4, generator=g), "audio_obs": torch.randn(n_ep, t_full, 1, 8, 4, generator=g),
           "vision_obs": torch.randn(n_ep, t_full, 1, 4, 4, generator=g)}
    for k, v in eps.items():
        out[f"ep/{k}"] = v.numpy()
    with tempfile.TemporaryDirectory() as td:
        lists = {}
        for k, v in eps.items():
            for i in range(n_ep):
                torch.save(v[i].clone(), Path(td) / f"{k}_{i:03d}.pt")
            lists[k] = sorted(Path(td).glob(f"{k}*"))

        def chain(noise: bool):  # noqa: ANN202
            fs = [tr.TakeFirstN(t_take)] + ([tr.GaussianNoise(0.1)] if noise else [])

            def f(x):  # noqa: ANN001, ANN202
                for t in fs:
                    x = t(x)
                return x
            return f

        order = ["act", "audio_obs", "vision_obs", "act", "audio_obs", "vision_obs"]
        transforms = [chain(True)] * 3 + [chain(False)] * 3
        stack = StackDataset(*[ds.EpisodeDataset(lists[k], t) for k, t in zip(order, transforms, strict=True)])
        torch.manual_seed(7)
        ref_batches = list(DataLoader(stack, batch_size=bs, shuffle=False, num_workers=0))
        torch.manual_seed(7)
        mine = list(ref_data.batches([lists[k] for k in order], transforms, bs))
    assert len(ref_batches) == len(mine) == 3
    for bi, (rb, mb) in enumerate(zip(ref_batches, mine, strict=True)):
        for fi, (r, m) in enumerate(zip(rb, mb, strict=True)):
            assert torch.equal(r, m), (bi, fi)
            out[f"batch/{bi}/{fi}"] = r.numpy()
    out["meta/batch"] = np.array([n_ep, t_full, t_take, bs])
    dst = ROOT / "tests" / "golden" / "data_feed.npz"
    np.savez_compressed(dst, **out)
    print("wrote", dst, f"{dst.stat().st_size / 1024:.1f} KB,", len(out), "arrays")


if __name__ == "__main__":
    main()
