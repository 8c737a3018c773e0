"""MFCC conditioning features on the GPU.

Replaces the CPU call in the reference's dataset (`data_loaders/gesture/data/dataset.py:81-95`):

    mfcc_vectors = mfcc(signal, winlen=0.06, winstep=(1/self.fps), samplerate=self.sr, numcep=27, nfft=5000)
    mfcc_vectors = (mfcc_vectors - self.mfcc_mean) / self.mfcc_std

(`python_speech_features.mfcc`, 26 filters, so 26 coefficients come back).  The tables below (DFT twiddles, mel
filterbank, DCT-II, lifter) are built once on the host in fp64 and kept on the device in fp32; the arithmetic runs in
libgdx (`gdx_mfcc`: framing kernel, DFT as a GEMM on the fp32 MFMA kernel, mel GEMM, cepstrum kernel).
python_speech_features is not installed in this environment, so agreement with the package itself is unpinned; the tests
compare against `oracle/mfcc.py`, a restatement of its published algorithm.
"""
import ctypes as C
import decimal
import math

import numpy as np
import torch

from .. import _lib
from ..engine import _ptr, _stream, f32c


def _round_half_up(x):
    return int(decimal.Decimal(x).quantize(decimal.Decimal("1"), rounding=decimal.ROUND_HALF_UP))


def _mel_filterbank(nfilt, nfft, samplerate, lowfreq=0.0, highfreq=None):
    highfreq = highfreq or samplerate / 2
    to_mel = lambda hz: 2595 * np.log10(1 + hz / 700.0)          # noqa: E731
    to_hz = lambda mel: 700 * (10 ** (mel / 2595.0) - 1)         # noqa: E731
    pts = np.linspace(to_mel(lowfreq), to_mel(highfreq), nfilt + 2)
    bins = np.floor((nfft + 1) * to_hz(pts) / samplerate)
    fb = np.zeros([nfilt, nfft // 2 + 1])
    for j in range(nfilt):
        for i in range(int(bins[j]), int(bins[j + 1])):
            fb[j, i] = (i - bins[j]) / (bins[j + 1] - bins[j])
        for i in range(int(bins[j + 1]), int(bins[j + 2])):
            fb[j, i] = (bins[j + 2] - i) / (bins[j + 2] - bins[j + 1])
    return fb


class MfccExtractor:
    """`extractor(signal)` -> [num_frames, 26] fp32 on the signal's device (the reference's `mfcc_vectors`)."""

    def __init__(self, device, sr=22050, fps=30, winlen=0.06, numcep=27, nfilt=26, nfft=5000, preemph=0.97, ceplifter=22,
                 mfcc_mean=None, mfcc_std=None):
        self.device = torch.device(device)
        self.sr, self.nfft, self.nfilt, self.preemph = sr, nfft, nfilt, float(preemph)
        self.numcep = min(numcep, nfilt)                          # dct(...)[:, :numcep] of nfilt columns
        self.frame_len = _round_half_up(winlen * sr)
        self.frame_step = _round_half_up((1 / fps) * sr)
        if self.frame_len > nfft:
            raise ValueError("frame length exceeds nfft (python_speech_features would truncate the frame)")
        nbins = nfft // 2 + 1
        self.Lp, self.nbp = -(-self.frame_len // 32) * 32, -(-nbins // 64) * 64
        i = np.arange(self.frame_len)[None, :]
        k = np.arange(nbins)[:, None]
        ang = 2.0 * np.pi * ((k * i) % nfft) / nfft
        dft = np.zeros((2 * self.nbp, self.Lp))
        dft[:nbins, : self.frame_len] = np.cos(ang)
        dft[self.nbp: self.nbp + nbins, : self.frame_len] = -np.sin(ang)
        mel = np.zeros((64, self.nbp))
        mel[:nfilt, :nbins] = _mel_filterbank(nfilt, nfft, sr)
        kk = np.arange(nfilt)[:, None]
        ii = np.arange(nfilt)[None, :]
        dct = np.cos(np.pi * kk * (2 * ii + 1) / (2 * nfilt)) * np.sqrt(2.0 / nfilt)
        dct[0] *= 1 / np.sqrt(2.0)
        lift = 1 + (ceplifter / 2.0) * np.sin(np.pi * np.arange(self.numcep) / ceplifter) if ceplifter > 0 else np.ones(self.numcep)
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(self.device)   # noqa: E731
        self.dft, self.mel, self.dct, self.lift = dev(dft), dev(mel), dev(dct[: self.numcep]), dev(lift)
        self.mean = dev(np.asarray(mfcc_mean)[: self.numcep]) if mfcc_mean is not None else None
        self.std = dev(np.asarray(mfcc_std)[: self.numcep]) if mfcc_std is not None else None

    def num_frames(self, n):
        return 1 if n <= self.frame_len else 1 + int(math.ceil((1.0 * n - self.frame_len) / self.frame_step))

    def __call__(self, signal):
        x = f32c(signal, "signal").reshape(-1)
        n = x.numel()
        F = self.num_frames(n)
        rows = F + _lib.GDX_ROW_PAD                # gdx.h: workspace rows per stage
        work = torch.zeros(rows * (self.Lp + 3 * self.nbp + 64) + F, device=x.device, dtype=torch.float32)
        out = torch.empty(F, self.numcep, device=x.device, dtype=torch.float32)
        lib = _lib.load()
        _lib.check(lib.gdx_mfcc(_ptr(x), n, self.frame_len, self.frame_step, F, self.nfft, self.nfilt, self.numcep,
                                C.c_float(self.preemph), _ptr(self.dft), _ptr(self.mel), _ptr(self.dct), _ptr(self.lift),
                                _ptr(self.mean) if self.mean is not None else None,
                                _ptr(self.std) if self.std is not None else None, _ptr(work), _ptr(out),
                                _stream(x.device)), lib)
        return out
