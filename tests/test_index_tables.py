"""Gather / scatter index tables: bit-exact against the values the reference's own walker produces on the
shipped levels (recorded in SURVEY.md section 8c from running mujoco_parent.py's __find_in_nested_dict with
xmltodict), and the level generator against the reference's data files when those are present."""
import os

import numpy as np
import pytest

from mjrl_amd import levels, mjcf, xmldict
from mjrl_amd.mujoco_parent import MuJoCoParent

REF = "/root/reference"
JOINTS_1 = ["root", "hip_1", "ankle_1", "hip_2", "ankle_2", "hip_3", "ankle_3", "hip_4", "ankle_4"]


def tables(level, agents, free_joint=False):
    p = MuJoCoParent.tables_only(levels.level_path(level), free_joint=free_joint)
    spaces = {a: (p.get_observation_space_mujoco(a), p.get_action_space_mujoco(a)) for a in agents}
    return p, spaces


def test_world_joint_order_is_depth_first():
    tree = xmldict.parse(open(levels.level_path("two_agent.xml")).read())
    world = xmldict.find_in_nested_dict(tree, parent="worldbody")
    names = [j["@name"] for j in xmldict.find_in_nested_dict(world, parent="joint")]
    assert names == JOINTS_1 + [n + "_2" for n in JOINTS_1]


def test_two_agent_tables_match_survey_golden_values():
    p, spaces = tables("two_agent.xml", ["sender", "receiver"])
    assert p.agents_action_index == {"sender": [2, 3, 4, 5, 6, 7, 0, 1], "receiver": [10, 11, 12, 13, 14, 15, 8, 9]}
    for agent, sens in (("sender", [0]), ("receiver", [1])):
        idx = p.agents_observation_index[agent]
        assert idx["sensors"] == sens
        assert idx["qpos"] == list(range(30)) and idx["qvel"] == list(range(28))   # every agent sees all joints
        obs_space, act_space = spaces[agent]
        assert len(obs_space["low"]) == 59                                   # 1 + nq + nv
        assert obs_space["low"][0] == -1 and obs_space["high"][0] == 20.0
        assert all(v == -np.inf for v in obs_space["low"][1:]) and all(v == np.inf for v in obs_space["high"][1:])
        assert act_space == {"low": [-1.0] * 8, "high": [1.0] * 8}


def test_three_sensor_level_tables():
    p, spaces = tables("two_agent_3sensors.xml", ["sender", "receiver"])
    assert p.agents_observation_index["sender"]["sensors"] == [0, 2, 4, 5, 6]
    assert p.agents_observation_index["receiver"]["sensors"] == [1, 3, 7, 8, 9]
    assert spaces["sender"][0]["low"][:5] == [-1, 0, -5.0, -5.0, -5.0]
    assert spaces["sender"][0]["high"][:5] == [20.0, 20.0, 5.0, 5.0, 5.0]


def test_single_agent_and_free_joint_tables():
    p, _ = tables("single_agent.xml", ["sender"])
    assert p.agents_action_index["sender"] == list(range(8))
    p, spaces = tables("two_agent.xml", ["sender", "receiver"], free_joint=True)
    assert p.agents_action_index == {"sender": [0, 1, 5], "receiver": [14, 15, 19]}
    assert spaces["sender"][1] == {"low": [-1, -1, -1], "high": [1, 1, 1]}


def test_framexaxis_level_tables():
    p, spaces = tables("sensor_framexaxis.xml", ["receiver"])
    assert p.agents_observation_index["receiver"]["sensors"] == [0, 1, 2]
    assert spaces["receiver"][0]["low"][:3] == [-1, -1, -1] and spaces["receiver"][0]["high"][:3] == [1, 1, 1]
    assert len(spaces["receiver"][0]["low"]) == 3 + 7 + 6


def test_model_sizes_match_the_survey():
    m = mjcf.compile_mjcf(levels.level_path("two_agent.xml"))
    assert (m.nbody, m.ngeom, m.njnt, m.nq, m.nv, m.nu, m.nsite, m.ncam) == (36, 35, 18, 30, 28, 16, 2, 2)
    assert sorted(np.bincount(m.geom_type).tolist(), reverse=True)[:4] == [24, 8, 2, 1]   # capsules, boxes, spheres, plane
    assert m.body_depth.max() == 4
    m3 = mjcf.compile_mjcf(levels.level_path("two_agent_3sensors.xml"))
    assert m3.nsensordata == 10
    m1 = mjcf.compile_mjcf(levels.level_path("single_agent.xml"))
    assert (m1.nbody, m1.ngeom, m1.nq, m1.nv, m1.nu, m1.nsensordata) == (21, 20, 15, 14, 8, 1)
    m4 = mjcf.compile_mjcf(levels.level_path("four_agent.xml"))
    assert (m4.nq, m4.nv, m4.nu, m4.ngeom) == (60, 56, 32, 61)


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")
@pytest.mark.parametrize("name", sorted(levels.REFERENCE_FILES))
def test_generated_level_compiles_like_the_reference_file(name):
    ours = mjcf.compile_mjcf(levels.level_path(name))
    theirs = mjcf.compile_mjcf(os.path.join(REF, levels.REFERENCE_FILES[name]))
    assert ours.names == theirs.names
    for key in theirs.arrays:
        assert np.array_equal(ours.arrays[key], theirs.arrays[key]), key
    # and the dict walk the table builders rely on sees the same joints, sites, sensors and motors
    a = xmldict.parse(open(levels.level_path(name)).read())
    b = xmldict.parse(open(os.path.join(REF, levels.REFERENCE_FILES[name])).read())
    for parent in ("joint", "site", "motor", "camera"):
        get = lambda d: [x.get("@name", x.get("@joint")) for x in xmldict.find_in_nested_dict(d, parent=parent)]
        assert get(a) == get(b), parent
    assert xmldict.find_in_nested_dict(a, parent="sensor") == xmldict.find_in_nested_dict(b, parent="sensor")


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")
def test_every_level_file_of_the_reference_tree_is_inside_the_supported_subset():
    """All MJCF files the reference ships (benchmarking/levels, Testing/levels, Testing/sensor_levels) compile -- nothing in
    them is refused by the loud subset check -- and fit the step kernel's limits (one wavefront: nv, bodies, joints <= 64,
    geoms <= 128, chains of at most 8 dofs); Testing/levels/Model2..10 are the 2-agent arena in other colours (the level
    variants of mujoco_parent.py:351-356)."""
    import glob
    files = sorted(glob.glob(os.path.join(REF, "**", "*.xml"), recursive=True))
    assert len(files) >= 19
    base = mjcf.compile_mjcf(os.path.join(REF, "Testing/levels/Model2.xml"))
    for path in files:
        m = mjcf.compile_mjcf(path)
        assert 1 <= m.nv <= 64 and m.nbody <= 64 and m.njnt <= 64 and m.ngeom <= 128 and m.maxdofdepth + 1 <= 8, path
        if "/Testing/levels/Model" in path and not path.endswith("Model1.xml"):
            for key in m.arrays:
                if key != "geom_rgba":
                    assert np.array_equal(m.arrays[key], base.arrays[key]), (path, key)


def test_blob_sections_are_aligned_to_their_element_size():
    """The float64 sections of a packed model start on 8-byte boundaries and the int32 sections on 4-byte ones (the
    header is 8 bytes + 4 bytes per size field, so the number of size fields has to be even): the device reads them with
    dwordx2 / dwordx4 loads and the CPU builds are sanitized for misaligned access."""
    import struct
    from mjrl_amd import blob, levels, mjcf
    assert len(blob.SIZE_FIELDS) % 2 == 0
    for level in ("two_agent.xml", "single_agent.xml", "sensor_touch.xml"):
        packed = blob.pack(mjcf.compile_mjcf(levels.level_path(level)))
        sizes = dict(zip(blob.SIZE_FIELDS, struct.unpack_from(f"<{len(blob.SIZE_FIELDS)}i", packed, 8)))
        off = 8 + 4 * len(blob.SIZE_FIELDS)
        assert off % 8 == 0
        off += 8 * len(blob.OPT_FIELDS)
        for name, expr in blob.F64_FIELDS:
            assert off % 8 == 0, name
            assert blob.section_range(packed, name)[0] == off
            off += 8 * eval(expr, {}, sizes)
        for name, expr in blob.I32_FIELDS:
            assert off % 4 == 0, name
            n = eval(expr, {}, sizes)
            off += 4 * (n + (n % 2))
        assert off == len(packed)
