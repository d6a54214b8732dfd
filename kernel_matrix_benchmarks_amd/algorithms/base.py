"""The plugin contract of kernel-matrix-benchmarks, restated for a tree that does
not contain the reference.

Reference: ``kernel_matrix_benchmarks/algorithms/base.py`` -- ``BaseAlgorithm``
(:7-48), ``BaseProduct`` (:51-116), ``BaseSolver`` (:119-167).  When the reference
package is importable (the plugin dropped into its tree) its own
classes are used, so ``isinstance`` checks against the reference's bases hold;
otherwise the equivalents below provide the same attributes, call order and
defaults.  The runner only relies on ``task`` and on the methods
(runner.py:70-148).
"""
import numpy as np

try:  # the reference tree, when this plugin is installed into it
    from kernel_matrix_benchmarks.algorithms.base import (  # type: ignore
        BaseAlgorithm,
        BaseProduct,
        BaseSolver,
    )

    USING_REFERENCE_BASES = True
except ImportError:
    USING_REFERENCE_BASES = False

    class BaseAlgorithm(object):
        """ctor kwargs: kernel, dimension, normalize_rows, precision (base.py:8-29)."""

        def __init__(self, *, kernel, dimension, normalize_rows=False, precision=np.float64):
            self.kernel = kernel
            self.dimension = dimension
            self.precision = precision
            self.normalize_rows = normalize_rows
            self.name = "BaseAlgorithm()"

        def done(self):
            """Release resources, also after an exception (runner.py:174-176)."""

        def get_memory_usage(self):
            """kB held by the process (base.py:35-38)."""
            import psutil

            return psutil.Process().memory_info().rss / 1024

        def set_query_arguments(self, **kwargs):
            """Parameters applied after fit() and before query() (runner.py:123)."""

        def get_additional(self):
            """Extra scalars stored as attributes of the result file (runner.py:162)."""
            return {}

        def __str__(self):
            return self.name

    class BaseProduct(BaseAlgorithm):
        task = "product"

        def prepare_data(self, *, source_points, target_points, same_points=False,
                         density_estimation=False):
            """Untimed: receives y (M,D) and x (N,D) as float64."""

        def fit(self):
            """Timed pre-computation."""

        def prepare_query(self, *, source_signal):
            """Untimed: receives b (M,E) as float64; called before every query run."""

        def query(self):
            """Timed; stores the result in self.res and returns None."""
            self.res = None

        def get_result(self):
            """Untimed: (N,E) float64, C-contiguous (base.py:107-116)."""
            return np.ascontiguousarray(self.res, dtype=np.float64)

    class BaseSolver(BaseAlgorithm):
        task = "solver"

        def prepare_data(self, *, source_points):
            """Untimed: receives y (M,D) as float64."""

        def fit(self):
            """Timed pre-computation."""

        def prepare_query(self, *, target_signal):
            """Untimed: receives a (N,E) as float64."""

        def query(self):
            raise NotImplementedError()

        def get_result(self):
            """Untimed: (M,E) float64, C-contiguous (base.py:158-167)."""
            return np.ascontiguousarray(self.res, dtype=np.float64)
