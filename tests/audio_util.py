"""Deterministic synthetic 16 kHz test audio (no sample files travel to the GPU box)."""
import numpy as np


def synth_speech(seconds, seed=0, sr=16000):
    """Speech-like signal: harmonic 'voiced' segments with moving formant-ish partials plus
    noise bursts, gated by a ~4 Hz syllable envelope, peak around 0.3."""
    rng = np.random.default_rng(seed)
    n = int(seconds * sr)
    t = np.arange(n) / sr
    f0 = 110 + 40 * np.sin(2 * np.pi * 0.7 * t + rng.uniform(0, 6)) + 15 * np.sin(2 * np.pi * 3.1 * t)
    phase = 2 * np.pi * np.cumsum(f0) / sr
    x = np.zeros(n)
    for h in range(1, 24):
        fc = h * f0
        amp = np.exp(-((fc - (600 + 300 * np.sin(2 * np.pi * 0.9 * t))) / 500.0) ** 2) + \
            0.5 * np.exp(-((fc - (1800 + 500 * np.sin(2 * np.pi * 0.5 * t + 1))) / 700.0) ** 2)
        x += amp * np.sin(h * phase + rng.uniform(0, 6))
    noise = rng.standard_normal(n)
    k = np.array([1.0, -0.8])
    noise = np.convolve(noise, k, mode="same")
    env = np.clip(np.sin(2 * np.pi * 4.0 * t + rng.uniform(0, 6)) * 1.5 + 0.3, 0, 1)
    gate = (np.sin(2 * np.pi * 0.37 * t + 2.0) > -0.6).astype(float)
    fric = (np.sin(2 * np.pi * 1.3 * t + 0.5) > 0.7).astype(float)
    y = (x * env * (1 - fric) + 0.6 * noise * fric * env) * gate
    y = y / (np.abs(y).max() + 1e-9) * 0.3
    # quantise like a 16-bit WAV would
    return (np.round(y * 32768.0) / 32768.0).astype(np.float32)
