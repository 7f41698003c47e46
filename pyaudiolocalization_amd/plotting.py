"""Correlation plots (reference plotting.py:7-48): host-side visualisation, outside the hot path."""
from __future__ import annotations

import numpy as np


def plot_correlation_heatmap(corr_matrix, mic_positions, title="Heatmap of peak correlations between microphone pairs",
                             show_plot=True, save_path=None):
    import matplotlib.pyplot as plt
    names = [f"Mic {i + 1}" for i in range(len(mic_positions))]
    fig, ax = plt.subplots(figsize=(8, 6))
    image = ax.imshow(corr_matrix, cmap="viridis")
    ax.set_xticks(np.arange(len(names)), labels=names, rotation=45, ha="right", rotation_mode="anchor")
    ax.set_yticks(np.arange(len(names)), labels=names)
    fig.colorbar(image, ax=ax).ax.set_ylabel("Peak Correlation", rotation=-90, va="bottom")
    ax.set_title(title)
    fig.tight_layout()
    if save_path:
        fig.savefig(save_path)
    if show_plot:
        plt.show()
    plt.close(fig)


def plot_correlation_3d(corr_data, mic_pairs, fs, title="3D Cross-Correlation Plots", show_plot=True, save_path=None):
    import matplotlib.pyplot as plt
    fig = plt.figure(figsize=(10, 8))
    ax = fig.add_subplot(111, projection="3d")
    for row, (corr, (i, j)) in enumerate(zip(corr_data, mic_pairs)):
        span = (len(corr) - 1) / fs
        ax.plot(np.linspace(-span, span, len(corr)), np.full(len(corr), row), corr, label=f"Mic {i + 1} - Mic {j + 1}")
    ax.set_xlabel("Lags (s)")
    ax.set_ylabel("Microphone Pairs")
    ax.set_zlabel("Correlation")
    ax.set_title(title)
    ax.legend()
    if save_path:
        fig.savefig(save_path)
    if show_plot:
        plt.show()
    plt.close(fig)
