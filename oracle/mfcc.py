"""Oracle: MFCC front end of the `y['mfcc']` conditioning (test infrastructure only).  PARITY UNPINNED.

Reference call site: `data_loaders/gesture/data/dataset.py:81-95` --
    mfcc(signal, winlen=0.06, winstep=1/fps, samplerate=sr, numcep=27, nfft=5000); (m - mfcc_mean) / mfcc_std
with `from python_speech_features import mfcc` (dataset.py:6; pinned as python-speech-features==0.6 in the reference's
environment).  That package is NOT installed in this image and is not vendored by the reference, and the reference holds
no MFCC fixtures, so this file restates the package's published algorithm (base.py / sigproc.py of release 0.6) and is
pinned by nothing but itself: parity of the GPU path with it is "unpinned" with respect to the reference.
"""
import decimal
import math

import numpy as np


def round_half_up(number):
    return int(decimal.Decimal(number).quantize(decimal.Decimal('1'), rounding=decimal.ROUND_HALF_UP))


def hz2mel(hz):
    return 2595 * np.log10(1 + hz / 700.)


def mel2hz(mel):
    return 700 * (10 ** (mel / 2595.0) - 1)


def get_filterbanks(nfilt=26, nfft=512, samplerate=16000, lowfreq=0, highfreq=None):
    highfreq = highfreq or samplerate / 2
    melpoints = np.linspace(hz2mel(lowfreq), hz2mel(highfreq), nfilt + 2)
    bins = np.floor((nfft + 1) * mel2hz(melpoints) / samplerate)
    fbank = np.zeros([nfilt, nfft // 2 + 1])
    for j in range(0, nfilt):
        for i in range(int(bins[j]), int(bins[j + 1])):
            fbank[j, i] = (i - bins[j]) / (bins[j + 1] - bins[j])
        for i in range(int(bins[j + 1]), int(bins[j + 2])):
            fbank[j, i] = (bins[j + 2] - i) / (bins[j + 2] - bins[j + 1])
    return fbank


def frame_geometry(slen, samplerate, winlen, winstep):
    frame_len = round_half_up(winlen * samplerate)
    frame_step = round_half_up(winstep * samplerate)
    numframes = 1 if slen <= frame_len else 1 + int(math.ceil((1.0 * slen - frame_len) / frame_step))
    return frame_len, frame_step, numframes


def dct_ortho_matrix(n):
    """scipy.fftpack.dct(type=2, norm='ortho') as a matrix: out = x @ D.T"""
    k = np.arange(n)[:, None]
    i = np.arange(n)[None, :]
    D = np.cos(np.pi * k * (2 * i + 1) / (2 * n)) * np.sqrt(2.0 / n)
    D[0] *= 1 / np.sqrt(2.0)
    return D


def mfcc(signal, samplerate=16000, winlen=0.025, winstep=0.01, numcep=13, nfilt=26, nfft=512, lowfreq=0, highfreq=None,
         preemph=0.97, ceplifter=22, appendEnergy=True):
    signal = np.asarray(signal, dtype=np.float64)
    signal = np.append(signal[0], signal[1:] - preemph * signal[:-1])                      # sigproc.preemphasis
    frame_len, frame_step, numframes = frame_geometry(len(signal), samplerate, winlen, winstep)
    padlen = int((numframes - 1) * frame_step + frame_len)                                 # sigproc.framesig (rect. window)
    padsignal = np.concatenate((signal, np.zeros((padlen - len(signal),))))
    idx = np.arange(frame_len)[None, :] + (np.arange(numframes) * frame_step)[:, None]
    frames = padsignal[idx]
    pspec = 1.0 / nfft * np.square(np.absolute(np.fft.rfft(frames, nfft)))                  # sigproc.powspec
    energy = np.sum(pspec, 1)
    energy = np.where(energy == 0, np.finfo(float).eps, energy)
    fb = get_filterbanks(nfilt, nfft, samplerate, lowfreq, highfreq)
    feat = np.dot(pspec, fb.T)
    feat = np.where(feat == 0, np.finfo(float).eps, feat)
    feat = np.log(feat)
    feat = np.dot(feat, dct_ortho_matrix(nfilt).T)[:, :numcep]
    if ceplifter > 0:                                                                       # lifter
        n = np.arange(feat.shape[1])
        feat = (1 + (ceplifter / 2.) * np.sin(np.pi * n / ceplifter)) * feat
    if appendEnergy:
        feat[:, 0] = np.log(energy)
    return feat


def genea_mfcc(signal, sr=22050, fps=30, mfcc_mean=None, mfcc_std=None):
    """dataset.py:90-94."""
    m = mfcc(signal, winlen=0.06, winstep=(1 / fps), samplerate=sr, numcep=27, nfft=5000)
    if mfcc_mean is not None:
        m = (m - mfcc_mean) / mfcc_std
    return m
