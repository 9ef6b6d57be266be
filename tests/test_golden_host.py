"""Host-side modules against vectors produced by the reference's own importable modules
(tests/golden/make_golden.py imports MuJoCo_Gym/sensor.py, helper.py, data_store.py) and against the known
answers the reference's tests hold (Testing/sensor_test.py:25-26,44-45,63-64,83-84; Testing/data_store_test.py)."""
import copy
import json
import os

import numpy as np
import pytest

from mjrl_amd import data_store, helper, sensor

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "reference_vectors.json")) as fh:
    GOLD = json.load(fh)


@pytest.mark.parametrize("case", sorted(GOLD["sensors"]))
def test_process_sensors_matches_reference(case):
    rec = GOLD["sensors"][case]
    indices = {i: dict(e) for i, e in enumerate(rec["input"])}
    for agent, expect in rec["agents"].items():
        idx, picked = sensor.process_sensors(copy.deepcopy(indices), [{"@name": agent + "_sensor"}])
        assert idx == expect["indices"]
        assert sensor.create_sensor_observation_space(picked) == expect["space"]


@pytest.mark.parametrize("stype", sorted(GOLD["bounds"]))
def test_bounds_of_every_sensor_type(stype):
    assert sensor.create_sensor_observation_space([{"type": stype, "cutoff": "7.5"}]) == GOLD["bounds"][stype]


def test_reference_test_known_answers():
    """The four assertions of Testing/sensor_test.py, for the sensor part of the space."""
    space = sensor.create_sensor_observation_space
    assert space([{"type": "touch", "cutoff": "20"}]) == {"low": [0], "high": [20.0]}
    assert space([{"type": "accelerometer", "cutoff": "5"}]) == {"low": [-5.0] * 3, "high": [5.0] * 3}
    assert space([{"type": "rangefinder", "cutoff": "10"}]) == {"low": [-1], "high": [10.0]}
    assert space([{"type": "frameyaxis"}]) == {"low": [-1] * 3, "high": [1] * 3}


def test_sensor_without_site_raises_like_the_reference():
    assert GOLD["missing_site"] == "KeyError:'site'"
    with pytest.raises(KeyError, match="site"):
        sensor.process_sensors({0: {"name": "x", "data": [0.0] * 3}}, [])


def test_mat2euler_matches_reference():
    for rec in GOLD["mat2euler"]:
        assert np.allclose(helper.mat2euler_scipy(np.array(rec["mat"])), rec["euler"], atol=1e-12)


def test_update_deep_matches_reference():
    rec = GOLD["update_deep"]
    assert helper.update_deep(copy.deepcopy(rec["old"]), copy.deepcopy(rec["new"])) == rec["result"]


def test_data_store_trace_matches_reference():
    store = data_store.DataStore(["agent1", "agent2"])
    trace = []
    store.set_agent("agent1")
    store["key1"] = "value1"
    trace.append(["pre_commit_read", store["key1"]])
    store.commit()
    trace.append(["post_commit_read", store["key1"]])
    store["key1"] = "value2"
    trace.append(["second_write_before_commit", store["key1"]])
    store.set_agent("agent2")
    store["key2"] = 5
    store.commit()
    trace.append(["subset_agent2", store.get_agent_subset("agent2")])
    trace.append(["repr", repr(store)])
    for label, fn in (("invalid_agent", lambda: store.set_agent("agent3")),
                      ("invalid_subset", lambda: store.get_agent_subset("agent3"))):
        with pytest.raises(ValueError) as exc:
            fn()
        trace.append([label, "ValueError:" + str(exc.value)])
    store.set_agent("global")
    with pytest.raises(ValueError) as exc:
        store["g"] = 1
    trace.append(["global_write", "ValueError:" + str(exc.value)])
    fresh = data_store.DataStore(["a"])
    with pytest.raises(ValueError) as exc:
        fresh["k"]
    trace.append(["no_agent_read", "ValueError:" + str(exc.value)])
    assert trace == GOLD["data_store_trace"]


# ---- the seven cases of Testing/data_store_test.py:13-98, restated
class TestDataStoreCases:
    def setup_method(self):
        self.store = data_store.DataStore(["agent1", "agent2"])

    def test_set_agent_valid(self):
        self.store.set_agent("agent1")
        assert self.store.current_agent == "agent1"

    def test_set_agent_invalid(self):
        with pytest.raises(ValueError):
            self.store.set_agent("agent3")

    def test_get_agent_subset_valid(self):
        assert self.store.get_agent_subset("agent1") == {}

    def test_get_agent_subset_invalid(self):
        with pytest.raises(ValueError):
            self.store.get_agent_subset("agent3")

    def test_setitem_is_buffered(self):
        self.store.set_agent("agent1")
        self.store["key1"] = "value1"
        assert self.store["key1"] is None

    def test_commit_publishes(self):
        self.store.set_agent("agent1")
        self.store["key1"] = "value1"
        self.store.commit()
        assert self.store["key1"] == "value1"

    def test_repr(self):
        self.store.set_agent("agent1")
        self.store["key1"] = "value1"
        self.store.commit()
        assert repr(self.store) == "{'agent1': {'key1': 'value1'}, 'agent2': {}}"
