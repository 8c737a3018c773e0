"""Batch assembly for the sampler's `y` dict: same names and results as the reference's `data_loaders/tensors.py:3-66`.

`gg_collate` turns the GENEA dataset's items -- tuples (motion [len, J], text, length, audio [n], mfcc [len, 26],
seed poses [n_seed, J]) -- into `(motion [B, J, 1, Tmax], {'y': {...}})` with the layouts `MDM.forward` expects: ragged
motions zero-padded to the longest one, `mask [B, 1, 1, Tmax]` true on the valid frames, `mfcc [B, 26, 1, T]`,
`seed [B, J, 1, n_seed]`, raw `audio [B, n]`, the captions as a list.  Host-side torch only (this runs in the DataLoader
workers, before anything reaches the device); pinned by the reference-generated fixture `tests/golden/collate.npz`.
"""
import torch


def lengths_to_mask(lengths, max_len):
    """[B] lengths -> bool [B, max_len], true where frame index < length (`tensors.py:3-6`)."""
    frame = torch.arange(max_len, device=lengths.device)
    return frame[None, :] < lengths[:, None]


def collate_tensors(batch):
    """Stack same-rank tensors of different sizes into one zero-padded tensor (`tensors.py:9-19`)."""
    rank = batch[0].dim()
    extent = [max(t.size(d) for t in batch) for d in range(rank)]
    out = batch[0].new_zeros((len(batch), *extent))
    for row, t in zip(out, batch):
        row[tuple(slice(0, n) for n in t.shape)] = t
    return out


def collate(batch):
    """List of item dicts ('inp' plus optional 'lengths', 'text', 'mfcc', 'audio', 'seed') -> (motion, {'y': ...})
    (`tensors.py:22-52`).  `None` items are dropped; without 'lengths' the length is the last extent of 'inp'."""
    items = [b for b in batch if b is not None]
    first = items[0]
    motion = collate_tensors([b["inp"] for b in items])
    if "lengths" in first:
        lengths = torch.as_tensor([b["lengths"] for b in items])
    else:
        lengths = torch.as_tensor([b["inp"].shape[-1] for b in items])
    y = {"mask": lengths_to_mask(lengths, motion.shape[-1])[:, None, None, :], "lengths": lengths}
    if "text" in first:
        y["text"] = [b["text"] for b in items]
    for key in ("mfcc", "audio"):                         # already carry a leading batch axis of 1
        if key in first:
            y[key] = torch.cat([b[key] for b in items], dim=0)
    if "seed" in first:
        y["seed"] = torch.stack([b["seed"] for b in items], dim=0)
    return motion, {"y": y}


def gg_collate(batch):
    """Adapter from the GENEA dataset's item tuples to `collate` (`tensors.py:55-66`): time goes last, a singleton
    feature axis is inserted, everything but the raw audio becomes fp32."""
    def to_j1t(a):                                        # [len, C] -> [C, 1, len]
        return torch.as_tensor(a).t().float().unsqueeze(1)

    return collate([{"inp": to_j1t(motion), "text": text, "lengths": length,
                     "audio": torch.as_tensor(audio).unsqueeze(0), "mfcc": to_j1t(mfcc).unsqueeze(0),
                     "seed": to_j1t(seed)}
                    for motion, text, length, audio, mfcc, seed in batch])
