"""ORACLE (test infrastructure only) -- CPU restatement of the reference's episode feed.

Follows ``multimodal_rssm/models/dataset.py`` and ``models/mrssm/dataset.py``:

* ``EpisodeDataset.__getitem__`` = ``transform(load_tensor(path))``                      (dataset.py:84-112)
* the six datasets are zipped by ``torch.utils.data.StackDataset``                        (mrssm/dataset.py:166-182)
* ``DataLoader(batch_size, shuffle=...)`` default-collates each field with ``torch.stack`` (dataset.py:347-386)
* 80/20 split of the SORTED path lists                                                    (dataset.py:67-81, mrssm 161-163)

Only ``tests/`` may import this module.  Pinned by ``tests/golden/data_feed.npz`` (``oracle/gen_golden_data.py`` runs the
reference's own ``transform.py`` / ``dataset.py`` code on the same tensors in the build container).
"""

from __future__ import annotations

from collections.abc import Callable, Iterator
from pathlib import Path

import torch
from torch import Tensor

Transform = Callable[[Tensor], Tensor]


def split_path_list(path_list: list[Path], train_ratio: float) -> tuple[list[Path], list[Path]]:
    split_point = int(len(path_list) * train_ratio)
    return path_list[:split_point], path_list[split_point:]


def normalize_observation_shape(observations: Tensor) -> Tensor:
    if observations.dim() == 5:  # noqa: PLR2004  (N,T,H,W,C) -> (N,T,C,H,W)
        return observations.permute(0, 1, 4, 2, 3)
    if observations.dim() == 4:  # noqa: PLR2004  (N,T,H,W) -> (N,T,1,H,W)
        return observations.unsqueeze(2)
    return observations


def batches(path_lists: list[list[Path]], transforms: list[Transform], batch_size: int) -> Iterator[tuple[Tensor, ...]]:
    """Unshuffled batches of the 6-tuple: field k = stack over the batch of ``transforms[k](load(path_lists[k][i]))``.

    Draw order of a random transform: the DataLoader iterator first draws one int64 from the global generator for its
    base seed (``torch.utils.data.dataloader._BaseDataLoaderIter.__init__``); a batch is then fetched through
    ``StackDataset.__getitems__``, which walks the datasets in tuple order and, inside each, the batch's indices --
    field-major, sample-minor."""
    torch.empty((), dtype=torch.int64).random_()
    n = len(path_lists[0])
    for lo in range(0, n, batch_size):
        rows = range(lo, min(lo + batch_size, n))
        yield tuple(torch.stack([t(torch.load(pl[i], weights_only=True)) for i in rows]) for pl, t in zip(path_lists, transforms, strict=True))
