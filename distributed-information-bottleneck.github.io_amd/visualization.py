"""Artefacts of the hot path's callbacks, mirroring reference visualization.py:
  save_compression_matrices (visualization.py:14-81) and save_distributed_info_plane (:83-113).
The sampling rule and the exp(-Bhattacharyya) matrix follow the reference (with SURVEY App. A
defects A2 fixed); the figure layout is this project's own.
"""
from __future__ import annotations

import os

import numpy as np

from . import utils

default_mpl_colors = ['#1f77b4', '#ff7f0e', '#2ca02c', '#d62728', '#9467bd', '#8c564b', '#e377c2', '#7f7f7f',
                      '#bcbd22', '#17becf']


def _plt():
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    return plt


def select_display_inputs(inp_features_raw, max_number_to_display=128, rng=None):
    """reference visualization.py:17-28: <10 unique raw values -> evaluate exactly those (histogram mode);
    otherwise `max_number_to_display` random rows, sorted by raw value.  Returns
    (indices into the feature rows, sorted raw values, value frequencies or None)."""
    raw = np.asarray(inp_features_raw)
    unique_vals, unique_idx = np.unique(raw, return_index=True)
    if len(unique_vals) < 10:
        counts = [np.average(raw == v) for v in unique_vals]
        return unique_idx[np.argsort(unique_vals)], np.sort(unique_vals), counts
    rng = rng or np.random
    sel = rng.choice(raw.shape[0], max_number_to_display)
    flat = raw[sel].reshape(len(sel), -1)[:, 0]
    order = np.argsort(flat)
    return sel[order], flat[order], None


def save_compression_matrices(feature_encoder, inp_features, out_fname, inp_features_raw=None, feature_label=None,
                              max_number_to_display=128, model=None, rng=None):
    """Encode a sample of feature values (deterministically), compute exp(-Bhattacharyya) between
    their Gaussians, save as PNG (if out_fname) and return the matrix."""
    inp_features = np.asarray(inp_features)
    if inp_features_raw is None:
        inp_features_raw = inp_features
    inds, sorted_raw, counts = select_display_inputs(inp_features_raw, max_number_to_display, rng)
    enc = np.asarray(feature_encoder(inp_features[inds]))
    emb_mus, emb_logvars = np.split(enc, 2, axis=-1)
    if model is not None and getattr(model, "_engine", None) is not None and hasattr(model._engine, "bhattacharyya"):
        bhat = model._engine.bhattacharyya(emb_mus, emb_logvars, emb_mus, emb_logvars).detach().cpu().numpy()
    else:
        bhat = utils.bhattacharyya_dist_mat(emb_mus, emb_logvars, emb_mus, emb_logvars)
    compression_matrix = np.exp(-bhat)
    if out_fname:
        plt = _plt()
        n = len(sorted_raw)
        fig = plt.figure(figsize=(6, 6))
        gs = fig.add_gridspec(2, 2, width_ratios=(1, 2), height_ratios=(1, 2), left=0.1, right=0.9, bottom=0.1,
                              top=0.9, wspace=0.05, hspace=0.05)
        ax = fig.add_subplot(gs[1, 1])
        ax.imshow(compression_matrix, vmin=0, vmax=1, cmap='Blues_r')
        ax.axis('off')
        axl = fig.add_subplot(gs[1, 0])
        axt = fig.add_subplot(gs[0, 1])
        if counts is not None:
            axl.barh(sorted_raw, counts, height=0.8)
            axt.bar(sorted_raw, counts, width=0.8)
            axl.set_xlim(0, 1)
            axt.set_ylim(0, 1)
        else:
            axl.plot(sorted_raw, np.arange(n), 'k', lw=3)
            axl.set_ylim(n, 0)
            axt.plot(np.arange(n), sorted_raw, 'k', lw=3)
            axt.set_xlim(0, n)
        for a in (axl, axt):
            a.set_xticks([]) if a is axt or counts is not None else None
            a.set_yticks([]) if a is axl and counts is None else None
            for s in ('right', 'top'):
                a.spines[s].set_visible(False)
        ax0 = fig.add_subplot(gs[0, 0])
        ax0.text(0, 0, feature_label if feature_label is not None else '')
        ax0.set_xlim(-0.5, 0.5)
        ax0.set_ylim(-0.5, 0.5)
        ax0.axis('off')
        fig.savefig(out_fname)
        plt.close(fig)
    return compression_matrix


def save_distributed_info_plane(kl_series, loss_series, outdir, entropy_y=None):
    """Distributed information plane (reference visualization.py:83-113): total KL (bits) on x,
    task loss on y (black), per-feature KL on a twin axis.  kl_series [epochs, F] in bits."""
    kl_series = np.asarray(kl_series)
    loss_series = np.asarray(loss_series)
    number_features = kl_series.shape[1]
    desired = min(1000, kl_series.shape[0])
    sieve = max(1, kl_series.shape[0] // desired)
    start = desired // 2
    parts = kl_series[::sieve]
    full = np.sum(parts, axis=-1)
    perf = loss_series[::sieve]
    lims = [0, 15]
    plt = _plt()
    fig = plt.figure(figsize=(8, 4))
    ax = fig.gca()
    ax.plot(full[start:], perf[start:], lw=4, color='k')
    if entropy_y is not None:
        ax.plot(lims, [entropy_y] * 2, 'k:')
    ax.set_xlim(lims)
    if number_features > 1:
        ax2 = ax.twinx()
        for f in range(number_features):
            ax2.plot(full[start:], parts[start:, f], color=default_mpl_colors[f % len(default_mpl_colors)], lw=4)
        ax.set_zorder(ax2.get_zorder() + 1)
        ax.patch.set_visible(False)
    os.makedirs(outdir, exist_ok=True)
    saveto = os.path.join(outdir, 'distributed_info_plane.png')
    fig.savefig(saveto, dpi=150)
    plt.close(fig)
    return saveto
