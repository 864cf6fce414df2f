"""AnchorGeneratorRange (mmdet/core/anchor/anchor3d_generator.py:44-79,105-128; unused by the shipped configs): layout and
end points.  Checked bit-exact against the reference function in the build container (with rets = list(np.meshgrid(..)),
the NumPy-2 form of its tuple assignment); here the defining properties."""
import numpy as np

import sassd  # noqa: F401
from sassd import anchors as A


def test_range_anchors():
    g = A.AnchorGeneratorRange([0, -40.0, -1.78, 70.4, 40.0, -1.78], sizes=[[1.6, 3.9, 1.56], [0.6, 0.8, 1.73]],
                               rotations=[0, 1.57])
    an = g([1, 200, 176])
    assert an.shape == (1, 200, 176, 2, 2, 7) and an.dtype == np.float32 and g.num_anchors_per_localization == 4
    assert np.array_equal(an[0, 0, :, 0, 0, 0], np.linspace(np.float32(0), np.float32(70.4), 176, dtype=np.float32))
    assert np.array_equal(an[0, :, 0, 0, 0, 1], np.linspace(np.float32(-40.0), np.float32(40.0), 200, dtype=np.float32))
    assert np.all(an[..., 2] == np.float32(-1.78))
    assert np.array_equal(an[0, 5, 7, 1, :, 3:6], np.float32([[0.6, 0.8, 1.73]] * 2))
    assert np.array_equal(an[0, 5, 7, 0, :, 6], np.float32([0, 1.57]))
    st = A.AnchorGeneratorStride(sizes=[1.6, 3.9, 1.56], anchor_strides=[0.4, 0.4, 1.0], anchor_offsets=[0.2, -39.8, -1.78],
                                 rotations=[0, 1.57])([1, 200, 176])
    assert st.shape == (1, 200, 176, 1, 2, 7) and abs(float(st[0, 0, 1, 0, 0, 0]) - 0.6) < 1e-6
