"""Level generator: writes the MJCF levels the benchmarks and tests run on.

``/root/reference`` does not travel to the GPU box, so the levels are produced here from their
parameters.  The arena and ant parameters are those of the reference's shipped data files
(benchmarking/levels/MultiAgentModel{,2Sensors,3Sensors}.xml, SingleAgentModel.xml,
Testing/sensor_levels/Model1-4.xml); tests/test_index_tables.py checks, whenever the reference tree is
present, that every generated level compiles to exactly the same tables as the file it stands for.
BASELINE.json names a ``MultiEnvs.xml`` that the reference does not ship;
``two_agent`` is the stand-in.  ``four_agent`` is the synthetic config-4 arena of SURVEY.md 8(d).
"""
from __future__ import annotations

import os
import xml.etree.ElementTree as ET

# static arena: (body name, body pos, box half sizes, rgba)
_ARENA = [
    ("", "0.01862761 -4.816084 0.5215917", "10 0.25 0.5", "0 .9 0 1", "border1_geom"),
    ("", "-0.02937651 4.738263 0.4082346", "10 0.25 0.5", "0 .9 0 1", "border2_geom"),
    ("", "9.789262 -0.04495001 0.4216571", "0.25 5 0.5", "0 .9 0 1", "border3_geom"),
    ("", "-9.848948 -0.07438982 0.4721622", "0.25 5 0.5", "0 .9 0 1", "border4_geom"),
    ("", "-0.7310539 -0.09776664 0.4394875", "0.25 5 0.5", "0 .9 0 1", "border5_geom"),
    ("choice_1", "7.02852 -2.071592 0.4710507", "1 1 0.5", "0 0 255 1", "choice_1_geom"),
    ("choice_2", "1.373154 -2.135309 0.3841939", "1 1 0.5", "255 0 0 1", "choice_2_geom"),
    ("reference", "-5.522553 -2.403258 0.4737153", "0.5 0.5 0.5", "0 0 255 1", "reference_geom"),
]

# ant legs: (leg body, aux geom, hip body, hip joint, leg geom, ankle joint, ankle geom, xy signs, ankle axis, ankle range)
_LEGS = [
    ("front_left_leg", "aux_1_geom", "aux_1", "hip_1", "left_leg_geom", "ankle_1", "left_ankle_geom", (1, 1), "-1 1 0", "30 70"),
    ("front_right_leg", "aux_2_geom", "aux_2", "hip_2", "right_leg_geom", "ankle_2", "right_ankle_geom", (-1, 1), "1 1 0", "-70 -30"),
    ("back_leg", "aux_3_geom", "aux_3", "hip_3", "back_leg_geom", "ankle_3", "third_ankle_geom", (-1, -1), "-1 1 0", "-70 -30"),
    ("right_back_leg", "aux_4_geom", "aux_4", "hip_4", "rightback_leg_geom", "ankle_4", "fourth_ankle_geom", (1, -1), "1 1 0", "30 70"),
]
_MOTOR_ORDER = ["hip_4", "ankle_4", "hip_1", "ankle_1", "hip_2", "ankle_2", "hip_3", "ankle_3"]


def _fmt(x):
    return f"{x:.1f}" if float(x) == int(x) else f"{x}"


def _ant(parent, name, pos, suffix, root_joint, site=True, cam_euler="90 180 0"):
    body = ET.SubElement(parent, "body", name=name, pos=pos)
    ET.SubElement(body, "camera", name=f"{name}_camera", pos="0 -0.6 0", euler=cam_euler)
    ET.SubElement(body, "geom", name=f"{name}_geom", pos="0 0 0", size="0.25", type="sphere")
    if site:
        ET.SubElement(body, "site", name=f"{name}_sensor", pos="1 0 0", size="0.01")
    ET.SubElement(body, "joint", armature="0", damping="0", limited="false", margin="0.01", name=root_joint,
                  pos="0 0 0", type="free")
    for leg, aux_geom, hip_body, hip, leg_geom, ankle, ankle_geom, (sx, sy), axis, rng in _LEGS:
        a, b = 0.2 * sx, 0.2 * sy
        leg_body = ET.SubElement(body, "body", name=leg + suffix, pos="0 0 0")
        ET.SubElement(leg_body, "geom", fromto=f"0.0 0.0 0.0 {_fmt(a)} {_fmt(b)} 0.0", name=aux_geom + suffix,
                      size="0.08", type="capsule")
        hb = ET.SubElement(leg_body, "body", name=hip_body + suffix, pos=f"{_fmt(a)} {_fmt(b)} 0")
        ET.SubElement(hb, "joint", axis="0 0 1", name=hip + suffix, pos="0.0 0.0 0.0", range="-30 30", type="hinge")
        ET.SubElement(hb, "geom", fromto=f"0.0 0.0 0.0 {_fmt(a)} {_fmt(b)} 0.0", name=leg_geom + suffix, size="0.08",
                      type="capsule")
        ab = ET.SubElement(hb, "body", pos=f"{_fmt(a)} {_fmt(b)} 0")
        ET.SubElement(ab, "joint", axis=axis, name=ankle + suffix, pos="0.0 0.0 0.0", range=rng, type="hinge")
        ET.SubElement(ab, "geom", fromto=f"0.0 0.0 0.0 {_fmt(2 * a)} {_fmt(2 * b)} 0.0", name=ankle_geom + suffix,
                      size="0.08", type="capsule")
    return body


def _arena(world, skip=()):
    ET.SubElement(world, "light", diffuse=".5 .5 .5", pos="0 0 3", dir="0 0 -1")
    floor = ET.SubElement(world, "body", pos="0 0 0", name="")
    ET.SubElement(floor, "geom", type="plane", size="10 5 1.110223E-16", euler="0 0 0", rgba="255 255 255 1", name="")
    for name, pos, size, rgba, gname in _ARENA:
        if name in skip and name:
            continue
        b = ET.SubElement(world, "body", pos=pos, name=name)
        ET.SubElement(b, "geom", type="box", size=size, euler="0 0 0", rgba=rgba, name=gname)


def _ant_defaults(root):
    default = ET.SubElement(root, "default")
    ET.SubElement(default, "joint", armature="1", damping="1", limited="true")
    ET.SubElement(default, "geom", density="5.0", friction="1 0.5 0.5", margin="0.01", rgba="0.8 0.6 0.4 1")


def _motors(root, suffixes, order=None):
    act = ET.SubElement(root, "actuator")
    for suffix in suffixes:
        for joint in (order or _MOTOR_ORDER):
            ET.SubElement(act, "motor", ctrllimited="true", ctrlrange="-1.0 1.0", joint=joint + suffix, gear="150")


def _sensors(root, agents, kinds):
    sensor = ET.SubElement(root, "sensor")
    cut = {"rangefinder": "20", "touch": "20", "accelerometer": "5"}
    for kind in kinds:
        for agent in agents:
            ET.SubElement(sensor, kind, name=f"{agent}_{kind}", site=f"{agent}_sensor", cutoff=cut[kind])


def _to_text(root):
    ET.indent(root, space="  ")
    return ET.tostring(root, encoding="unicode") + "\n"


def two_agent(sensors=("rangefinder",)) -> str:
    """2-agent ant arena = benchmarking/levels/MultiAgentModel.xml (1, 2 or 3 sensor kinds per agent)."""
    root = ET.Element("mujoco")
    _ant_defaults(root)
    world = ET.SubElement(root, "worldbody")
    _arena(world)
    _ant(world, "sender", "-5.522553 0.9194446 1", "", "root")
    _ant(world, "receiver", "4.595446 1.222577 1", "_2", "root_2")
    _sensors(root, ["sender", "receiver"], sensors)
    _motors(root, ["", "_2"])
    return _to_text(root)


def single_agent() -> str:
    """1-agent ant arena = benchmarking/levels/SingleAgentModel.xml."""
    root = ET.Element("mujoco")
    _ant_defaults(root)
    world = ET.SubElement(root, "worldbody")
    _arena(world, skip=("choice_1", "choice_2"))
    _ant(world, "sender", "-5.522553 0.9194446 1", "", "root")
    _sensors(root, ["sender"], ("rangefinder",))
    _motors(root, [""], order=["hip_1", "ankle_1", "hip_2", "ankle_2", "hip_3", "ankle_3", "hip_4", "ankle_4"])
    return _to_text(root)


def four_agent() -> str:
    """Synthetic 4-agent contact-heavy arena: the 2-agent arena with the ant
    subtree instantiated four times; no such level ships with the reference."""
    root = ET.Element("mujoco")
    _ant_defaults(root)
    world = ET.SubElement(root, "worldbody")
    _arena(world)
    spawn = [("sender", "-5.522553 0.9194446 1", "", "root"), ("receiver", "4.595446 1.222577 1", "_2", "root_2"),
             ("agent_3", "-3.5 1.6 1", "_3", "root_3"), ("agent_4", "3.0 2.4 1", "_4", "root_4")]
    for name, pos, suffix, joint in spawn:
        _ant(world, name, pos, suffix, joint)
    _sensors(root, [s[0] for s in spawn], ("rangefinder",))
    _motors(root, [s[2] for s in spawn])
    return _to_text(root)


def ant() -> str:
    """The single free ant on an unbounded floor = benchmarking/levels/Ant.xml: the level with the Runge-Kutta integrator
    (RK4, timestep 0.01), leg geoms that collide with the floor only (conaffinity 0 against the floor's 1), a tracking
    camera on the torso."""
    root = ET.Element("mujoco", model="ant")
    ET.SubElement(root, "compiler", angle="degree", coordinate="local", inertiafromgeom="true")
    ET.SubElement(root, "option", integrator="RK4", timestep="0.01")
    default = ET.SubElement(root, "default")
    ET.SubElement(default, "joint", armature="1", damping="1", limited="true")
    ET.SubElement(default, "geom", conaffinity="0", condim="3", density="5.0", friction="1 0.5 0.5", margin="0.01",
                  rgba="0.8 0.6 0.4 1")
    # (the floor's material, Ant.xml:15: its specular / shininess enter the fixed-function shading; its checker texture
    # and reflectance are outside the ray caster's image model)
    asset = ET.SubElement(root, "asset")
    ET.SubElement(asset, "material", name="MatPlane", reflectance="0.5", shininess="1", specular="1")
    world = ET.SubElement(root, "worldbody")
    ET.SubElement(world, "light", cutoff="100", diffuse="1 1 1", dir="-0 0 -1.3", directional="true", exponent="1",
                  pos="0 0 1.3", specular=".1 .1 .1")
    ET.SubElement(world, "geom", conaffinity="1", condim="3", material="MatPlane", name="floor", pos="0 0 0",
                  rgba="0.8 0.9 0.8 1", size="40 40 40", type="plane")
    body = ET.SubElement(world, "body", name="torso", pos="0 0 0.75")
    ET.SubElement(body, "camera", name="track", mode="trackcom", pos="0 -3 0.3", xyaxes="1 0 0 0 0 1")
    ET.SubElement(body, "geom", name="torso_geom", pos="0 0 0", size="0.25", type="sphere")
    ET.SubElement(body, "joint", armature="0", damping="0", limited="false", margin="0.01", name="root", pos="0 0 0",
                  type="free")
    for leg, aux_geom, hip_body, hip, leg_geom, ankle, ankle_geom, (sx, sy), axis, rng in _LEGS:
        a, b = 0.2 * sx, 0.2 * sy
        leg_body = ET.SubElement(body, "body", name=leg, pos="0 0 0")
        ET.SubElement(leg_body, "geom", fromto=f"0.0 0.0 0.0 {_fmt(a)} {_fmt(b)} 0.0", name=aux_geom, size="0.08", type="capsule")
        hb = ET.SubElement(leg_body, "body", name=hip_body, pos=f"{_fmt(a)} {_fmt(b)} 0")
        ET.SubElement(hb, "joint", axis="0 0 1", name=hip, pos="0.0 0.0 0.0", range="-30 30", type="hinge")
        ET.SubElement(hb, "geom", fromto=f"0.0 0.0 0.0 {_fmt(a)} {_fmt(b)} 0.0", name=leg_geom, size="0.08", type="capsule")
        ab = ET.SubElement(hb, "body", pos=f"{_fmt(a)} {_fmt(b)} 0")
        ET.SubElement(ab, "joint", axis=axis, name=ankle, pos="0.0 0.0 0.0", range=rng, type="hinge")
        ET.SubElement(ab, "geom", fromto=f"0.0 0.0 0.0 {_fmt(2 * a)} {_fmt(2 * b)} 0.0", name=ankle_geom, size="0.08", type="capsule")
    actuator = ET.SubElement(root, "actuator")
    for joint in _MOTOR_ORDER:
        ET.SubElement(actuator, "motor", ctrllimited="true", ctrlrange="-1.0 1.0", joint=joint, gear="150")
    return _to_text(root)


def sensor_level(kind: str) -> str:
    """Free box with one site sensor = Testing/sensor_levels/Model1-4.xml
    (touch / accelerometer / rangefinder / framexaxis)."""
    root = ET.Element("mujoco")
    world = ET.SubElement(root, "worldbody")
    _arena(world)
    body = ET.SubElement(world, "body", pos="4.595446 1.222577 0.4743838", euler="180 0 0", name="receiver")
    ET.SubElement(body, "geom", type="box", size="0.5 0.5 0.5", euler="0 0 0", rgba="255 255 0 1", name="receiver_geom")
    ET.SubElement(body, "joint", type="free", name="receiver_freeJoint")
    ET.SubElement(body, "camera", name="receiver_camera", pos="0 -0.6 0", euler="90 0 0")
    ET.SubElement(body, "site", name="receiver_sensor", pos="1 0 0", size="0.01")
    sensor = ET.SubElement(root, "sensor")
    if kind == "touch":
        ET.SubElement(sensor, "touch", name="receiver_touch", site="receiver_sensor", cutoff="20")
    elif kind == "accelerometer":
        ET.SubElement(sensor, "accelerometer", name="receiver_accelerometer", site="receiver_sensor", cutoff="5")
    elif kind == "rangefinder":
        ET.SubElement(sensor, "rangefinder", name="receiver_rangefinder", site="receiver_sensor", cutoff="10")
    elif kind == "framexaxis":
        ET.SubElement(sensor, "framexaxis", name="receiver_framexaxis", objtype="site", objname="receiver_sensor")
    else:
        raise ValueError(kind)
    return _to_text(root)


LEVELS = {
    "two_agent.xml": lambda: two_agent(("rangefinder",)),
    "two_agent_2sensors.xml": lambda: two_agent(("rangefinder", "touch")),
    "two_agent_3sensors.xml": lambda: two_agent(("rangefinder", "touch", "accelerometer")),
    "single_agent.xml": single_agent,
    "four_agent.xml": four_agent,
    "ant.xml": ant,
    "sensor_touch.xml": lambda: sensor_level("touch"),
    "sensor_accelerometer.xml": lambda: sensor_level("accelerometer"),
    "sensor_rangefinder.xml": lambda: sensor_level("rangefinder"),
    "sensor_framexaxis.xml": lambda: sensor_level("framexaxis"),
}

# which reference file each generated level stands for (checked by tests/test_index_tables.py::test_generated_level_compiles_like_the_reference_file when the reference tree is present)
REFERENCE_FILES = {
    "two_agent.xml": "benchmarking/levels/MultiAgentModel.xml",
    "two_agent_2sensors.xml": "benchmarking/levels/MultiAgentModel2Sensors.xml",
    "two_agent_3sensors.xml": "benchmarking/levels/MultiAgentModel3Sensors.xml",
    "single_agent.xml": "benchmarking/levels/SingleAgentModel.xml",
    "ant.xml": "benchmarking/levels/Ant.xml",
    "sensor_touch.xml": "Testing/sensor_levels/Model1.xml",
    "sensor_accelerometer.xml": "Testing/sensor_levels/Model2.xml",
    "sensor_rangefinder.xml": "Testing/sensor_levels/Model3.xml",
    "sensor_framexaxis.xml": "Testing/sensor_levels/Model4.xml",
}

_CACHE_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "levels")


def level_path(name: str) -> str:
    """Path of a generated level file (written on first use)."""
    os.makedirs(_CACHE_DIR, exist_ok=True)
    path = os.path.join(_CACHE_DIR, name)
    text = LEVELS[name]()
    if not os.path.exists(path) or open(path).read() != text:
        # (atomic: several ranks of a multi-GPU run may get here at once, and a reader must never see half a file)
        tmp = f"{path}.{os.getpid()}.tmp"
        with open(tmp, "w") as fh:
            fh.write(text)
        os.replace(tmp, path)
    return path


def write_all() -> list:
    return [level_path(name) for name in LEVELS]
