"""Episode feed: transforms, split and batch assembly against fixtures generated from the
reference's own ``transform.py`` / ``dataset.py`` (``oracle/gen_golden_data.py``), host logic of the DataModule on CPU,
and -- on the MI355X -- the fused gather kernel against the same fixtures (bit-exact) and at full size."""

from __future__ import annotations

from pathlib import Path

import numpy as np
import pytest
import torch

from multimodal_mtrssm_amd import dataset as ds
from multimodal_mtrssm_amd import transform as tr
from oracle import ref_data

GOLDEN = np.load(Path(__file__).parent / "golden" / "data_feed.npz")


def _t(key: str) -> torch.Tensor:
    return torch.from_numpy(GOLDEN[key])


def _chain(n: int, std: float | None) -> tr.Compose:
    return tr.Compose([tr.TakeFirstN(n)] + ([tr.GaussianNoise(std)] if std is not None else []))


def _config(root: Path, n: int = 6, std: float | None = 0.1, batch_size: int = 2, **kw) -> ds.EpisodeDataModuleConfig:  # noqa: ANN003
    ident = torch.nn.Identity()
    base = dict(data_name="toy", batch_size=batch_size, num_workers=0, gdrive_url="", action_preprocess=ident,
                action_input_transform=_chain(n, std), action_target_transform=_chain(n, None),
                audio_observation_file_name="audio.npy", vision_observation_file_name="vision.npy",
                audio_observation_preprocess=ident, vision_observation_preprocess=ident,
                audio_observation_input_transform=_chain(n, std), audio_observation_target_transform=_chain(n, None),
                vision_observation_input_transform=_chain(n, std), vision_observation_target_transform=_chain(n, None),
                data_root=root)
    base.update(kw)
    return ds.EpisodeDataModuleConfig(**base)


def _write_golden_episodes(d: Path) -> dict[str, list[Path]]:
    d.mkdir(parents=True, exist_ok=True)
    lists = {}
    for k in ("act", "audio_obs", "vision_obs"):
        eps = _t(f"ep/{k}")
        for i in range(eps.shape[0]):
            torch.save(eps[i].clone(), d / f"{k}_{i:03d}.pt")
        lists[k] = sorted(d.glob(f"{k}*"))
    return lists


# ---------------------------------------------------------------------------------------------
# CPU: transforms, split, restatement
# ---------------------------------------------------------------------------------------------
def test_transforms_match_the_reference_fixtures() -> None:
    vis, aud, act = _t("in/vision"), _t("in/audio"), _t("in/action")
    assert torch.equal(tr.NormalizeVisionImage()(vis), _t("out/normalize_vision"))
    assert torch.equal(tr.NormalizeAudioMelSpectrogram()(aud), _t("out/normalize_audio_default"))
    assert torch.equal(tr.NormalizeAudioMelSpectrogram(min_value=-80.0, max_value=0.0)(aud), _t("out/normalize_audio_yaml"))
    assert torch.equal(tr.TakeFirstN(5)(act), _t("out/take_first_5"))
    assert torch.equal(tr.RemoveDim(1, [0, 4])(act), _t("out/remove_dim"))
    torch.manual_seed(99)
    assert torch.equal(tr.GaussianNoise(0.1)(act), _t("out/gaussian_noise"))
    before = vis.clone()
    tr.NormalizeVisionImage()(vis)
    assert torch.equal(vis, before)  # works on a copy, as the reference does


def test_shape_normaliser_and_split_match_the_reference_fixtures() -> None:
    x5 = torch.arange(2 * 3 * 4 * 5 * 2.0).reshape(2, 3, 4, 5, 2)
    x4 = torch.arange(2 * 3 * 4 * 5.0).reshape(2, 3, 4, 5)
    assert torch.equal(ds.normalize_observation_shape(x5), _t("out/norm_shape_5d"))
    assert torch.equal(ds.normalize_observation_shape(x4), _t("out/norm_shape_4d"))
    assert torch.equal(ref_data.normalize_observation_shape(x5), _t("out/norm_shape_5d"))
    for n in (1, 4, 5, 10, 11):
        a, b = ds.split_path_list([Path(f"p{i}") for i in range(n)], 0.8)
        assert [len(a), len(b)] == GOLDEN[f"out/split_{n}"].tolist()
        assert a + b == [Path(f"p{i}") for i in range(n)]


def test_oracle_batches_match_the_reference_dataloader(tmp_path: Path) -> None:
    lists = _write_golden_episodes(tmp_path)
    n_ep, _, t_take, bs = GOLDEN["meta/batch"].tolist()
    order = ["act", "audio_obs", "vision_obs"] * 2
    transforms = [_chain(t_take, 0.1)] * 3 + [_chain(t_take, None)] * 3
    torch.manual_seed(7)
    got = list(ref_data.batches([lists[k] for k in order], transforms, bs))
    assert len(got) == (n_ep + bs - 1) // bs
    for bi, b in enumerate(got):
        for fi, f in enumerate(b):
            assert torch.equal(f, _t(f"batch/{bi}/{fi}")), (bi, fi)


def test_fused_chain_recognition() -> None:
    assert tr.fused_chain(_chain(30, 0.1)) == (30, 0.1)
    assert tr.fused_chain(_chain(30, None)) == (30, None)
    assert tr.fused_chain(torch.nn.Identity()) == (None, None)
    assert tr.fused_chain(tr.GaussianNoise(0.2)) == (None, 0.2)
    assert tr.fused_chain(tr.Compose([tr.GaussianNoise(0.1), tr.TakeFirstN(3)])) is None  # other order: not the fused kernel
    assert tr.fused_chain(tr.Compose([tr.TakeFirstN(3), tr.RemoveDim(1, [0])])) is None


def test_datamodule_host_logic_on_cpu(tmp_path: Path) -> None:
    """prepare_data from the raw arrays (mrssm/dataset.py:107-134), the effective-directory rule, the 80/20 split, and
    the refusal to assemble batches without the GPU."""
    from multimodal_mtrssm_amd._lib import MtrssmLibraryError

    g = torch.Generator().manual_seed(3)
    n, t = 10, 7
    raw = tmp_path / "toy"
    raw.mkdir()
    np.save(raw / "audio.npy", torch.rand(n, t, 8, 4, generator=g).numpy())          # (N,T,H,W)
    np.save(raw / "vision.npy", (torch.rand(n, t, 4, 4, 3, generator=g) * 255).numpy())  # (N,T,H,W,C)
    np.save(raw / "joint_states.npy", torch.randn(n, t, 4, generator=g).numpy())
    cfg = _config(tmp_path, n=5, batch_size=3, vision_observation_preprocess=tr.NormalizeVisionImage())
    dm = ds.EpisodeDataModule(cfg, device="cpu")
    dm.prepare_data()
    proc = tmp_path / "processed_toy"
    assert len(list(proc.glob("act_*.pt"))) == len(list(proc.glob("audio_obs_*.pt"))) == len(list(proc.glob("vision_obs_*.pt"))) == n
    v0 = torch.load(proc / "vision_obs_000.pt", weights_only=True)
    assert v0.shape == (t, 3, 4, 4) and float(v0.min()) >= -1.0 and float(v0.max()) <= 1.0
    assert torch.load(proc / "audio_obs_003.pt", weights_only=True).shape == (t, 1, 8, 4)
    dm.prepare_data()  # second call: processed data is ready, nothing to do
    dm.setup("fit")
    assert dm.train_streams[0].store.shape == (8, t, 4) and dm.val_streams[2].store.shape == (2, t, 3, 4, 4)
    assert len(dm.train_dataloader()) == 3 and len(dm.val_dataloader()) == 1
    with pytest.raises(MtrssmLibraryError):  # no CPU path for batch assembly
        next(iter(dm.val_dataloader()))
    # data/processed_data wins when it holds every kind (dataset.py:140-163)
    common = tmp_path / "processed_data"
    common.mkdir()
    assert cfg.get_effective_processed_data_dir(cfg.get_observation_glob_patterns()) == proc
    for k in ("act", "audio_obs", "vision_obs"):
        torch.save(torch.zeros(2, 4), common / f"{k}_000.pt")
    assert cfg.get_effective_processed_data_dir(cfg.get_observation_glob_patterns()) == common
    with pytest.raises(RuntimeError, match="setup"):
        ds.EpisodeDataModule(cfg, device="cpu").train_dataloader()
    with pytest.raises(FileNotFoundError, match="Google Drive"):
        ds.EpisodeDataModule(_config(tmp_path / "nowhere"), device="cpu").prepare_data()
    with pytest.raises(ValueError, match="Unknown file extension"):
        ds.load_tensor(tmp_path / "x.txt")


def test_data_parallel_loader_shards_are_disjoint_and_equal() -> None:
    """ADVICE r1 (dataset.py:168): with shuffle=True every rank must cut the SAME epoch permutation, run the same number
    of steps and hold equally many rows per step, also when n is not a multiple of batch_size * world (n=10, bs=4, world=4:
    an unequal tail would leave a rank without rows and deadlock the per-step all-reduce)."""
    for n, bs, world in ((10, 4, 4), (13, 6, 2), (7, 8, 3), (12, 4, 2)):
        ident = tr.Compose([])
        streams = tuple(ds._Stream(torch.arange(n, dtype=torch.float32).reshape(n, 1, 1).expand(n, 3, w).contiguous(), ident, ident)  # noqa: SLF001
                        for w in (4, 2, 2))
        loaders = [ds.DeviceEpisodeLoader(streams, bs, shuffle=True, rank=r, world=world, seed=11) for r in range(world)]
        for epoch in range(2):
            per_rank = [list(ld.index_batches()) for ld in loaders]
            steps = {len(b) for b in per_rank}
            assert steps == {(n + bs - 1) // bs}, (n, bs, world, steps)
            seen: list[int] = []
            for step in range(len(per_rank[0])):
                sizes = {int(b[step].numel()) for b in per_rank}
                assert len(sizes) == 1 and sizes.pop() > 0  # equal, non-empty shards: the all-reduce average is unbiased
                rows = torch.cat([b[step] for b in per_rank]).tolist()
                seen += rows
            # every episode is visited; duplicates only from the wrap-around padding of a ragged batch
            assert set(seen) == set(range(n))
            pad = sum((-min(bs, n - lo)) % world for lo in range(0, n, bs))
            assert len(seen) == n + pad
        # the next epoch is another permutation, again identical across ranks
        a = torch.cat(list(loaders[0].index_batches()))
        loaders[0].set_epoch(0)
        b = torch.cat(list(loaders[0].index_batches()))
        assert not torch.equal(a, b) or n <= 2
    one = ds.DeviceEpisodeLoader(streams, 5, shuffle=False)
    assert [int(b.numel()) for b in one.index_batches()] == [5, 5, 2]  # single rank keeps the short tail (reference DataLoader)
    # ADVICE r2 (dataset.py:189): a rank's rows are the CONTIGUOUS block FlatDataParallel.shard cuts and GlobalRowNoise.draw keys its
    # uniforms by -- the ranks' blocks, in rank order, ARE the one-rank batch (row g of the global batch meets the same noise)
    streams12 = tuple(ds._Stream(torch.arange(12, dtype=torch.float32).reshape(12, 1, 1).expand(12, 3, w).contiguous(), tr.Compose([]), tr.Compose([]))  # noqa: SLF001
                      for w in (4, 2, 2))
    whole = list(ds.DeviceEpisodeLoader(streams12, 4, shuffle=True, seed=3).index_batches())
    halves = [list(ds.DeviceEpisodeLoader(streams12, 4, shuffle=True, rank=r, world=2, seed=3).index_batches()) for r in range(2)]
    for step, rows in enumerate(whole):
        assert torch.equal(torch.cat([halves[0][step], halves[1][step]]), rows)


# ---------------------------------------------------------------------------------------------
# MI355X: the fused gather kernel
# ---------------------------------------------------------------------------------------------
def _replayed_noise(seed: int, batches: list[int], t_take: int, events: list[tuple[int, ...]]) -> list[list[torch.Tensor]]:
    """The standard normals the reference DataLoader run drew (oracle/ref_data.py: base-seed int64, then field-major)."""
    torch.manual_seed(seed)
    torch.empty((), dtype=torch.int64).random_()
    return [[torch.stack([torch.randn(t_take, *ev) for _ in range(b)]) for ev in events] for b in batches]


@pytest.mark.gpu
def test_gather_kernel_reproduces_the_reference_batches(tmp_path: Path) -> None:
    lists = _write_golden_episodes(tmp_path / "processed_toy")
    n_ep, t_full, t_take, bs = GOLDEN["meta/batch"].tolist()
    cfg = _config(tmp_path, n=t_take, std=0.1, batch_size=bs)
    dm = ds.EpisodeDataModule(cfg, device="cuda:0")
    streams = tuple(ds._Stream(dm._stack(lists[k]), _chain(t_take, 0.1), _chain(t_take, None)) for k in ("act", "audio_obs", "vision_obs"))  # noqa: SLF001
    loader = ds.DeviceEpisodeLoader(streams, bs, shuffle=False)
    sizes = [min(bs, n_ep - lo) for lo in range(0, n_ep, bs)]
    noise = _replayed_noise(7, sizes, t_take, [s.event_shape for s in streams])
    for bi, lo in enumerate(range(0, n_ep, bs)):
        idx = torch.arange(lo, lo + sizes[bi], device="cuda:0")
        got = loader.batch(idx, tuple(n.to("cuda:0") for n in noise[bi]))
        for fi, f in enumerate(got):
            assert torch.equal(f.cpu(), _t(f"batch/{bi}/{fi}")), (bi, fi)  # bit-exact, inputs included
    # the module's own split: 4 training episodes, 1 validation episode, targets = the stored prefix
    dm.setup("fit")
    assert dm.train_streams[0].store.shape[0] == 4 and dm.val_streams[0].store.shape[0] == 1
    (val,) = list(dm.val_dataloader())
    assert torch.equal(val[4].cpu(), _t("ep/audio_obs")[4:, :t_take]) and val[1].shape == val[4].shape
    assert 0.05 < float((val[1] - val[4]).std()) < 0.2  # device-drawn noise of std 0.1


@pytest.mark.gpu
def test_full_size_feed_properties() -> None:
    """BASELINE frame sizes (vision 1x64x64, audio 1x128x32, action 4), B=64, T=50 of 60 stored steps."""
    dev = "cuda:0"
    g = torch.Generator().manual_seed(5)
    n, t_full, t = 96, 60, 50
    stores = [torch.randn(n, t_full, 4, generator=g), torch.randn(n, t_full, 1, 128, 32, generator=g), torch.randn(n, t_full, 1, 64, 64, generator=g)]
    streams = tuple(ds._Stream(s.to(dev), _chain(t, 0.1), _chain(t, None)) for s in stores)  # noqa: SLF001
    loader = ds.DeviceEpisodeLoader(streams, 64, shuffle=True)
    idx = torch.randperm(n, generator=g)[:64]
    noise = tuple(torch.randn(64, t, *s.event_shape, generator=g) for s in streams)
    batch = loader.batch(idx.to(dev), tuple(x.to(dev) for x in noise))
    for k in range(3):
        want_t = stores[k][idx, :t]
        assert torch.equal(batch[3 + k].cpu(), want_t)
        assert torch.equal(batch[k].cpu(), want_t + noise[k] * 0.1)  # mul then add, each rounded: torch's expression
    # one epoch visits every episode exactly once; two ranks take disjoint halves of every global batch
    seen = torch.cat([b[3][:, 0, 0] for b in loader]).cpu()
    assert sorted(seen.tolist()) == sorted(stores[0][:, 0, 0].tolist())
    halves = [ds.DeviceEpisodeLoader(streams, 64, shuffle=False, rank=r, world=2) for r in range(2)]
    rows = [torch.cat([b[3][:, 0, 0] for b in h]).cpu() for h in halves]
    assert rows[0].numel() + rows[1].numel() == n and not set(rows[0].tolist()) & set(rows[1].tolist())
    # an action width that is not a multiple of 4 floats takes the generic (torch, still on-device) route
    odd = ds._Stream(torch.randn(8, 9, 7, generator=g).to(dev), _chain(5, None), _chain(5, None))  # noqa: SLF001
    i, tgt = odd.batch(torch.tensor([3, 1], device=dev), None)
    assert torch.equal(i, tgt) and tgt.shape == (2, 5, 7)
