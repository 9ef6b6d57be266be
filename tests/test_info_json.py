"""Info-JSON object tags (/root/reference MuJoCo_Gym/mujoco_rl.py:93-112, 355-395): the host helpers
``filter_by_tag`` / ``get_data`` and the device form of a tag.  The fixture tests/golden/two_agent_info.json follows the
schema the reference's code implies ({"environment": {"objects": {name: {"tags": [...], ...}}}, "areas": {area:
{"objects": {...}}}}); the reference ships no example file.  The physics side of ``get_data`` is served here by a stub over
the CPU oracle (no GPU in this suite); the -m gpu tests run the same calls on the device."""
import json
import os

import numpy as np
import pytest

from mjrl_amd import blob, dynamics, levels, mjcf
from mjrl_amd.helper import mat2euler_scipy
from mjrl_amd.mujoco_parent import ModelView, MuJoCoParent, _Named
from mjrl_amd.mujoco_rl import MuJoCoRL
from oracle.oracle import OracleEnv

HERE = os.path.dirname(os.path.abspath(__file__))
INFO = os.path.join(HERE, "golden", "two_agent_info.json")


class OracleData:
    """``env.data`` over one oracle copy: what DataView serves from the device."""

    def __init__(self, model, ora):
        self.m, self.o = model, ora

    def body(self, name):
        b = self.m.name2id("body", name)
        return _Named(id=b, name=name, xpos=self.o.xpos[b], xipos=self.o.xipos[b], xmat=self.o.xmat[b])

    def geom(self, name):
        g = self.m.name2id("geom", name)
        return _Named(id=g, name=name, xpos=self.o.geom_xpos[g], xmat=self.o.geom_xmat[g])


@pytest.fixture()
def env():
    model = mjcf.compile_mjcf(levels.level_path("two_agent.xml"))
    ora = OracleEnv(blob.pack(model))
    ora.step(5)
    e = object.__new__(MuJoCoRL)
    e.agents, e.n_env, e._compiled, e.model = ["sender", "receiver"], 1, model, ModelView(model)
    e.data = OracleData(model, ora)
    e.xml_path, e.xml_paths, e.info_jsons = levels.level_path("two_agent.xml"), levels.level_path("two_agent.xml"), INFO
    e._load_info_json()
    return e, model, ora


def test_info_json_is_loaded_like_the_reference_does(env):
    e, model, ora = env
    assert e.info_name_list == ["reference", "choice_1", "border1_geom", "border2_geom", "sender"]
    assert e.info_json["areas"]["area_south"]["objects"]["choice_2"]["colour"] == "red"
    # a list of JSON files is matched to the level by file stem (mujoco_rl.py:96-103)
    e.xml_paths, e.info_jsons = [e.xml_path], ["/nowhere/other.json", INFO.replace("two_agent_info", "two_agent")]
    with pytest.raises(Exception):
        e._load_info_json()                     # two_agent.json does not exist -> open() fails like in the reference
    e.info_jsons = [INFO, INFO]
    with pytest.raises(Exception, match="Length mismatch"):
        e._load_info_json()


def test_filter_by_tag_visits_environment_objects_then_areas(env):
    e, model, ora = env
    assert e.tagged_names("target") == ["reference", "choice_1", "choice_2"]
    assert e.tagged_names("wall") == ["border1_geom", "border3_geom"]
    assert e.tagged_names("south") == ["choice_2"] and e.tagged_names("nothing") == []
    found = e.filter_by_tag("target")
    assert [d["name"] for d in found] == ["reference", "choice_1", "choice_2"] and all(d["type"] == "body" for d in found)
    b = model.name2id("body", "choice_2")
    assert np.array_equal(found[2]["position"], ora.xipos[b])
    walls = e.filter_by_tag("wall")
    assert [d["type"] for d in walls] == ["geom", "geom"] and walls[0]["id"] == model.name2id("geom", "border1_geom")
    e.info_json = {"environment": {"objects": {}}}
    with pytest.raises(KeyError):
        e.filter_by_tag("target")               # the reference indexes info_json["areas"] unconditionally (mujoco_rl.py:371)


def test_get_data_merges_the_json_attributes_but_keeps_the_physics(env):
    e, model, ora = env
    data = e.get_data("reference")
    b = model.name2id("body", "reference")
    # JSON keys are merged except position / orientation / mass, which stay the simulator's (mujoco_rl.py:391-394)
    assert data["class"] == "Cube" and data["colour"] == "blue" and data["tags"] == ["target", "landmark"]
    assert np.array_equal(data["position"], ora.xipos[b]) and data["mass"][0] == model.body_mass[b] != 99
    assert np.allclose(data["orientation"], mat2euler_scipy(ora.xmat[b]))
    assert data["type"] == "body" and data["id"] == b and data["name"] == "reference"
    geom = e.get_data("border1_geom")
    g = model.name2id("geom", "border1_geom")
    assert geom["type"] == "geom" and geom["class"] == "Border" and np.array_equal(geom["position"], ora.geom_xpos[g])
    assert np.array_equal(geom["color"], model.geom_rgba[g]) and geom["shape"][0] == mjcf.GEOM_BOX
    plain = e.get_data("receiver")              # not in the JSON: the physics record alone
    assert set(plain) == {"position", "mass", "orientation", "id", "name", "type"}
    # objects listed under an area are not merged (only environment.objects is consulted, mujoco_rl.py:390)
    assert "colour" not in e.get_data("choice_2")
    with pytest.raises(KeyError):
        e.get_data("no_such_object")


def test_device_form_of_a_tag(env):
    e, model, ora = env
    names = model.names
    assert e.tag_refs("target") == [(0, names["body"].index(n)) for n in ("reference", "choice_1", "choice_2")]
    assert e.tag_refs("wall") == [(1, names["geom"].index("border1_geom")), (1, names["geom"].index("border3_geom"))]
    e.info_json["environment"]["objects"]["ghost"] = {"tags": ["target"]}
    with pytest.raises(KeyError):
        e.tag_refs("target")


def test_counter_based_choices_are_reproducible_and_spread():
    z = dynamics.mix64(7, np.arange(4096), 1, 33, 0)
    assert z.dtype == np.uint64 and len(np.unique(z)) == 4096
    assert np.array_equal(z, dynamics.mix64(7, np.arange(4096), 1, 33, 0))
    assert int(dynamics.mix64(0, 0, 0, 0, 0)) == 0 and int(dynamics.mix64(1, 0, 0, 0, 0)) != 0
    picks = dynamics.pick_of(z, 3)
    assert set(picks) == {0, 1, 2} and abs(np.bincount(picks) - 4096 / 3).max() < 150
    assert not np.array_equal(picks, dynamics.pick_of(dynamics.mix64(7, np.arange(4096), 1, 33, 1), 3))


def test_levels_that_differ_in_colours_only_are_one_model(tmp_path):
    """Testing/levels/Model2-10.xml differ in the boxes' rgba only.  Such an xmlPath list is one model
    plus colour variants (blob.same_physics); a transparent geom is not a colour change (rangefinders ignore it), and a
    moved wall is another model."""
    text = open(levels.level_path("two_agent.xml")).read()
    base = blob.pack(mjcf.compile_mjcf_string(text))
    recoloured = blob.pack(mjcf.compile_mjcf_string(text.replace('rgba="0 .9 0 1"', 'rgba="0.9 0 0.2 1"')))
    transparent = blob.pack(mjcf.compile_mjcf_string(text.replace('rgba="0 .9 0 1" name="border1_geom"', 'rgba="0 .9 0 0" name="border1_geom"')))
    moved = blob.pack(mjcf.compile_mjcf_string(text.replace('pos="7.02852 -2.071592 0.4710507"', 'pos="6.5 -2.071592 0.4710507"')))
    assert base != recoloured and blob.same_physics(base, recoloured)
    assert not blob.same_physics(base, transparent) and not blob.same_physics(base, moved)
    assert not blob.same_physics(base, blob.pack(mjcf.compile_mjcf(levels.level_path("two_agent_2sensors.xml"))))
    lo, hi = blob.section_range(base, "geom_rgba")
    assert hi - lo == 8 * 4 * 35 and base[:lo] == recoloured[:lo] and base[hi:] == recoloured[hi:]
    # the reference's own variants, when the tree is there
    ref = "/root/reference/Testing/levels"
    if os.path.isdir(ref):
        packed = [blob.pack(mjcf.compile_mjcf(os.path.join(ref, f"Model{k}.xml"))) for k in (2, 3, 7, 10)]
        assert all(blob.same_physics(packed[0], other) for other in packed[1:])
        assert len({bytes(p) for p in packed}) > 1
