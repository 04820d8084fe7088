"""
Module name of the reference (/root/reference/localmd/diagnostic_plots.py) for its image routines:
``from localmd_amd.diagnostic_plots import make_correlation_image`` works like the reference's import.  The routines
themselves live in localmd_amd/diagnostic_images.py (device kernels); the plotly figure builders of the reference file
(make_pmd_corr_diagnostic_plot, make_pmd_component_graph, plot_pmd_components, construct_index) are presentation code
outside the hot path and are not provided.
"""
from .diagnostic_images import (  # noqa: F401
    make_autocorrelation_image,
    make_correlation_image,
    make_pmd_correlation_image,
    make_residual_correlation_image,
)

__all__ = ["make_correlation_image", "make_autocorrelation_image", "make_pmd_correlation_image", "make_residual_correlation_image"]
