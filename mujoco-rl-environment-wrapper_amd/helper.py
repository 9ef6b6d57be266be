"""Host helpers with the semantics of the reference's MuJoCo_Gym/helper.py.

``mat2euler_scipy`` (helper.py:6-18) feeds ``get_data()["orientation"]``; ``update_deep``
(helper.py:21-31) is what ``MuJoCoRL.reset`` uses to merge data-store copies.
"""
from __future__ import annotations

import numpy as np
from scipy.spatial.transform import Rotation


def mat2euler_scipy(mat) -> np.ndarray:
    """Rotation matrix (9 values or 3x3) -> intrinsic z-y-x Euler angles in degrees."""
    rot = Rotation.from_matrix(np.asarray(mat, dtype=np.float64).reshape(3, 3))
    z, y, x = rot.as_euler("zyx", degrees=True)
    return np.array([z, y, x])


def update_deep(old_dict: dict, new_dict: dict) -> dict:
    """Merge ``new_dict`` into ``old_dict`` in place; nested dicts merge, everything else overwrites."""
    for key, value in new_dict.items():
        target = old_dict.get(key)
        if isinstance(target, dict) and isinstance(value, dict):
            update_deep(target, value)
        else:
            old_dict[key] = value
    return old_dict
