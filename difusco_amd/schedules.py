"""Host-side diffusion tables and the inference time schedule (numpy float64, computed once).

Mirrors the public surface of the reference's ``difusco/utils/diffusion_schedulers.py`` - class
names, constructor arguments and attribute names (``T``, ``beta``, ``Q_bar``, ``alpha``,
``alphabar``) - so code written against the reference keeps working.  Only what the inference
path reads is provided (no ``sample``: forward noising is training-only).
"""
import math

import numpy as np


def _betas(T: int, schedule: str) -> np.ndarray:
    # diffusion_schedulers.py:16-23 / :53-60
    if schedule == "linear":
        return np.linspace(1e-4, 2e-2, T)
    if schedule == "cosine":
        s = 0.008
        grid = np.arange(0, T + 1, 1)
        f = np.cos(math.pi * 0.5 * (grid / T + s) / (1 + s)) ** 2
        f0 = np.cos(math.pi * 0.5 * (0 / T + s) / (1 + s)) ** 2
        abar = f / f0
        return np.clip(1 - (abar[1:] / abar[:-1]), None, 0.999)
    raise ValueError(f"Unknown diffusion schedule: {schedule}")


class CategoricalDiffusion(object):
    """Cumulative 2-state transition matrices Q_bar[t], t = 0..T (diffusion_schedulers.py:46-72)."""

    def __init__(self, T, schedule):
        self.T = T
        self.beta = _betas(T, schedule)
        I2, J2 = np.eye(2), np.ones((2, 2))
        self.Qs = (1 - self.beta)[:, None, None] * I2[None] + (self.beta / 2)[:, None, None] * J2[None]
        acc = np.eye(2)
        stack = [acc]
        for Q in self.Qs:
            acc = acc @ Q
            stack.append(acc)
        self.Q_bar = np.stack(stack, axis=0)

    def posterior_constants(self, t: int, target_t: int) -> np.ndarray:
        """The four fp32 scalars of the categorical posterior (pl_meta_model.py:115-137):
        p(x_s = 1 | x_t = b, p0, p1) = c0[b]*p0 + c1[b]*p1, returned as [c0[0], c0[1], c1[0], c1[1]].
        Computed with the reference's own rounding sequence: float64 inverse/product, cast to fp32,
        then fp32 multiply and divide."""
        Qt = (np.linalg.inv(self.Q_bar[target_t]) @ self.Q_bar[t]).astype(np.float32)
        Qb_t = self.Q_bar[t].astype(np.float32)
        Qb_s = self.Q_bar[target_t].astype(np.float32)
        c0 = [np.float32(Qt[1, b] * Qb_s[0, 1]) / Qb_t[0, b] for b in (0, 1)]
        c1 = [np.float32(Qt[1, b] * Qb_s[1, 1]) / Qb_t[1, b] for b in (0, 1)]
        return np.array(c0 + c1, dtype=np.float32)


class GaussianDiffusion(object):
    """alphabar[t] = prod_{s<=t} alpha[s], alpha[0] = 1 (diffusion_schedulers.py:9-28)."""

    def __init__(self, T, schedule):
        self.T = T
        self.beta = _betas(T, schedule)
        self.betabar = np.cumprod(self.beta)
        self.alpha = np.concatenate((np.array([1.0]), 1 - self.beta))
        self.alphabar = np.cumprod(self.alpha)

    def posterior_constants(self, t: int, target_t: int, inference_trick) -> np.ndarray:
        """[a, b, c, d, branch] with x_s = a*(x_t - b*eps) + c*eps (DDIM, branch 0) or
        a*(x_t - b*eps) + d*z (DDPM, branch 1).  pl_meta_model.py:158-174."""
        atbar, atbar_target = self.alphabar[t], self.alphabar[target_t]
        if inference_trick is None or t <= 1:
            at = self.alpha[t]
            atbar_prev = self.alphabar[t - 1]
            beta_tilde = self.beta[t - 1] * (1 - atbar_prev) / (1 - atbar)
            return np.array([1 / np.sqrt(at), (1 - at) / np.sqrt(1 - atbar), 0.0, np.sqrt(beta_tilde), 1.0],
                            dtype=np.float32)
        if inference_trick == "ddim":
            return np.array([np.sqrt(atbar_target / atbar), np.sqrt(1 - atbar), np.sqrt(1 - atbar_target), 0.0, 0.0],
                            dtype=np.float32)
        raise ValueError("Unknown inference trick {}".format(inference_trick))


class InferenceSchedule(object):
    """Step i of ``inference_T`` -> (t1, t2) on the T-step training grid (diffusion_schedulers.py:85-111)."""

    def __init__(self, inference_schedule="linear", T=1000, inference_T=1000):
        self.inference_schedule = inference_schedule
        self.T = T
        self.inference_T = inference_T

    def _at(self, frac: float) -> int:
        if self.inference_schedule == "linear":
            return self.T - int(frac * self.T)
        if self.inference_schedule == "cosine":
            return self.T - int(np.sin(frac * np.pi / 2) * self.T)
        raise ValueError("Unknown inference schedule: {}".format(self.inference_schedule))

    def __call__(self, i):
        assert 0 <= i < self.inference_T
        t1 = np.clip(self._at(float(i) / self.inference_T), 1, self.T)
        t2 = np.clip(self._at(float(i + 1) / self.inference_T), 0, self.T - 1)
        return t1, t2
