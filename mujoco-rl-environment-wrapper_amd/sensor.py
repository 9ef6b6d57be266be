"""Sensor address map and per-type observation bounds -- semantics of the reference's
MuJoCo_Gym/sensor.py:1-116 (known answers: Testing/sensor_test.py:25-26,44-45,63-64,83-84).

``process_sensors`` walks the sensors in sensor-id order, gives each a run of ``sensordata`` addresses as
long as its ``data`` vector, and keeps those whose site belongs to the agent.  A sensor dict without a
``"site"`` key raises ``KeyError`` exactly like the reference.
"""
from __future__ import annotations

_CUTOFF_TYPES = ("rangefinder", "touch", "accelerometer")

# type -> (count, low, high); "c" stands for the sensor's cutoff
_BOUNDS = {}
for _t in ("touch", "actuatorpos", "clock"):
    _BOUNDS[_t] = (1, 0, "c")
for _t in ("accelerometer", "velocimeter", "gyro", "force", "torque", "magnetometer", "framepos", "ballangvel",
           "framelinvel", "frameangvel", "framelinacc", "frameangacc"):
    _BOUNDS[_t] = (3, "-c", "c")
_BOUNDS["rangefinder"] = (1, -1, "c")
for _t in ("jointlimitpos", "jointlimitvel", "jointlimitfrc", "tendonlimitpos", "tendonlimitvel", "tendonlimitfrc"):
    _BOUNDS[_t] = (1, "-c", 0)
_BOUNDS["camprojection"] = (2, 0, "c")
for _t in ("ballquat", "framequat"):
    _BOUNDS[_t] = (4, "-c", "c")
for _t in ("framexaxis", "frameyaxis", "framezaxis"):
    _BOUNDS[_t] = (3, -1, 1)
for _t in ("subtreecom", "subtreelinvel", "subtreeangmom", "jointpos", "jointvel", "tendonpos", "tendonvel",
           "actuatorvel", "actuatorfrc", "jointactuatorfrc"):
    _BOUNDS[_t] = (1, "-c", "c")
_BOUNDS["user"] = (1, -1, 1)
_BOUNDS["plugin"] = (1, 0, 100)


def process_sensor(sensor: dict, index: int):
    width = len(sensor["data"])
    info = {"indices": list(range(index, index + width)), "site": sensor["site"], "type": sensor["type"]}
    if info["type"] in _CUTOFF_TYPES:
        info["cutoff"] = sensor["cutoff"]
    return info, index + width


def extract_agent_indices(new_indices: dict, agent_sites: list):
    site_names = [site["@name"] for site in agent_sites]
    agent_sensors = [info for info in new_indices.values() if info["site"] in site_names]
    agent_indices = [i for info in agent_sensors for i in info["indices"]]
    return agent_indices, agent_sensors


def process_sensors(indices: dict, agent_sites: list):
    by_name, address = {}, 0
    for _, sensor in sorted(indices.items()):
        by_name[sensor["name"]], address = process_sensor(sensor, address)
    return extract_agent_indices(by_name, agent_sites)


def create_sensor_observation_space(agent_sensors: list) -> dict:
    space = {"low": [], "high": []}
    for sensor in agent_sensors:
        rule = _BOUNDS.get(sensor["type"])
        if rule is None:
            continue
        count, low, high = rule

        def value(v):
            if v == "c":
                return float(sensor["cutoff"])
            if v == "-c":
                return -1 * float(sensor["cutoff"])
            return v
        for _ in range(count):
            space["low"].append(value(low))
            space["high"].append(value(high))
    return space
