"""Generates tests/golden/reference_vectors.json by importing the reference's own pure-Python modules
(MuJoCo_Gym/sensor.py, helper.py, data_store.py import fine in the build container; mujoco_parent.py and
mujoco_rl.py do not, because mujoco / gymnasium / xmltodict are not installed -- SURVEY.md section 8c).

Run in the build container only (it reads /root/reference):  python tests/golden/make_golden.py
The JSON holds inputs and the reference's outputs; no reference source text is stored.
"""
import importlib.util
import json
import os
import sys

import numpy as np

REF = "/root/reference/MuJoCo_Gym"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_vectors.json")


def load(name):
    spec = importlib.util.spec_from_file_location("ref_" + name, os.path.join(REF, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    sensor, helper, data_store = load("sensor"), load("helper"), load("data_store")
    out = {}

    # --- sensor.process_sensors / create_sensor_observation_space on the shipped sensor sets
    def sensor_set(entries):
        return {i: dict(e) for i, e in enumerate(entries)}
    three = [
        dict(name="sender_rangefinder", data=[0.0], site="sender_sensor", type="rangefinder", cutoff="20"),
        dict(name="receiver_rangefinder", data=[0.0], site="receiver_sensor", type="rangefinder", cutoff="20"),
        dict(name="sender_touch", data=[0.0], site="sender_sensor", type="touch", cutoff="20"),
        dict(name="receiver_touch", data=[0.0], site="receiver_sensor", type="touch", cutoff="20"),
        dict(name="sender_accelerometer", data=[0.0] * 3, site="sender_sensor", type="accelerometer", cutoff="5"),
        dict(name="receiver_accelerometer", data=[0.0] * 3, site="receiver_sensor", type="accelerometer", cutoff="5"),
    ]
    cases = {"three_sensors": three, "two_sensors": three[:4], "one_sensor": three[:2],
             "framexaxis": [dict(name="receiver_framexaxis", data=[0.0] * 3, site="receiver_sensor", type="frameyaxis")],
             "touch_level": [dict(name="receiver_touch", data=[0.0], site="receiver_sensor", type="touch", cutoff="20")],
             "accel_level": [dict(name="receiver_accelerometer", data=[0.0] * 3, site="receiver_sensor", type="accelerometer", cutoff="5")],
             "range_level": [dict(name="receiver_rangefinder", data=[0.0], site="receiver_sensor", type="rangefinder", cutoff="10")]}
    out["sensors"] = {}
    for key, entries in cases.items():
        rec = {"input": entries, "agents": {}}
        for agent in ("sender", "receiver"):
            idx, picked = sensor.process_sensors(sensor_set(entries), [{"@name": agent + "_sensor"}])
            rec["agents"][agent] = {"indices": idx, "space": sensor.create_sensor_observation_space(picked)}
        out["sensors"][key] = rec
    # every type the bounds table knows
    types = ["touch", "actuatorpos", "clock", "accelerometer", "velocimeter", "gyro", "force", "torque", "magnetometer",
             "framepos", "ballangvel", "framelinvel", "frameangvel", "framelinacc", "frameangacc", "rangefinder",
             "jointlimitpos", "jointlimitvel", "jointlimitfrc", "tendonlimitpos", "tendonlimitvel", "tendonlimitfrc",
             "camprojection", "ballquat", "framequat", "framexaxis", "frameyaxis", "framezaxis", "subtreecom",
             "subtreelinvel", "subtreeangmom", "jointpos", "jointvel", "tendonpos", "tendonvel", "actuatorvel",
             "actuatorfrc", "jointactuatorfrc", "user", "plugin", "unknown_type"]
    out["bounds"] = {t: sensor.create_sensor_observation_space([{"type": t, "cutoff": "7.5"}]) for t in types}
    # a sensor without a site key
    try:
        sensor.process_sensors({0: {"name": "x", "data": [0.0] * 3}}, [])
        out["missing_site"] = "no error"
    except KeyError as exc:
        out["missing_site"] = "KeyError:" + str(exc)

    # --- helper
    rng = np.random.default_rng(5)
    mats = [np.eye(3), np.array([[0.0, -1, 0], [1, 0, 0], [0, 0, 1]])]
    for _ in range(6):
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        w, x, y, z = q
        mats.append(np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                              [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                              [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]]))
    out["mat2euler"] = [{"mat": m.reshape(-1).tolist(), "euler": helper.mat2euler_scipy(m.reshape(-1)).tolist()} for m in mats]
    a, b = {"a": {"x": 1}, "b": 2, "c": {"d": {"e": 1}}}, {"a": {"y": 3}, "b": {"z": 1}, "c": {"d": {"f": 2}}}
    import copy
    out["update_deep"] = {"old": copy.deepcopy(a), "new": copy.deepcopy(b), "result": helper.update_deep(a, b)}

    # --- DataStore behaviour trace
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
        try:
            fn()
            trace.append([label, "no error"])
        except ValueError as exc:
            trace.append([label, "ValueError:" + str(exc)])
    store.set_agent("global")
    try:
        store["g"] = 1
        trace.append(["global_write", "no error"])
    except ValueError as exc:
        trace.append(["global_write", "ValueError:" + str(exc)])
    fresh = data_store.DataStore(["a"])
    try:
        fresh["k"]
        trace.append(["no_agent_read", "no error"])
    except ValueError as exc:
        trace.append(["no_agent_read", "ValueError:" + str(exc)])
    out["data_store_trace"] = trace

    with open(OUT, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print("wrote", OUT)


if __name__ == "__main__":
    sys.exit(main())
