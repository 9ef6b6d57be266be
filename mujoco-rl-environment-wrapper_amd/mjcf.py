"""MJCF subset compiler: level XML -> flat constant tables for the batched stepper.

The reference hands its XML to ``mujoco.MjModel.from_xml_path`` (mujoco_parent.py:126).  That
package is not part of the reference tree; this module re-implements the compile semantics
the 19 shipped levels rely on from MuJoCo's published XML reference:
default classes, degrees, intrinsic xyz euler, ``fromto`` capsules, inertia inferred from geoms,
depth-first body numbering with static bodies kept, weld-based collision filtering, motor
actuators, and the constants derived at ``qpos0`` (``body_invweight0``, ``dof_invweight0``,
``meaninertia``).

Everything here is host-side, run once per level.  Output is a :class:`Model` of numpy arrays
that ``blob.pack`` serialises for the C-ABI (include/mjrl.h) and for the oracle.
"""
from __future__ import annotations

import math
import xml.etree.ElementTree as ET
from dataclasses import dataclass, field

import numpy as np

# enums (values follow MuJoCo's public mjtGeom / mjtJoint / mjtSensor ordering where it matters:
# collision functions are indexed with the lower geom type first)
GEOM_PLANE, GEOM_HFIELD, GEOM_SPHERE, GEOM_CAPSULE, GEOM_ELLIPSOID, GEOM_CYLINDER, GEOM_BOX = range(7)
GEOM_TYPES = {"plane": GEOM_PLANE, "sphere": GEOM_SPHERE, "capsule": GEOM_CAPSULE, "box": GEOM_BOX}
JNT_FREE, JNT_BALL, JNT_SLIDE, JNT_HINGE = range(4)
JNT_TYPES = {"free": JNT_FREE, "hinge": JNT_HINGE, "slide": JNT_SLIDE}
SENS_TOUCH, SENS_ACCELEROMETER, SENS_RANGEFINDER, SENS_FRAMEXAXIS, SENS_FRAMEYAXIS, SENS_FRAMEZAXIS = range(6)
SENSOR_TYPES = {
    "touch": (SENS_TOUCH, 1), "accelerometer": (SENS_ACCELEROMETER, 3),
    "rangefinder": (SENS_RANGEFINDER, 1), "framexaxis": (SENS_FRAMEXAXIS, 3),
    "frameyaxis": (SENS_FRAMEYAXIS, 3), "framezaxis": (SENS_FRAMEZAXIS, 3),
}
INT_EULER, INT_RK4 = 0, 1

MINVAL = 1e-15


# ----------------------------------------------------------------------------- small math
def _vec(text, n=None, default=None):
    if text is None:
        return None if default is None else np.array(default, dtype=np.float64)
    v = np.array([float(t) for t in text.split()], dtype=np.float64)
    if n is not None and v.size < n and default is not None:
        full = np.array(default, dtype=np.float64)
        full[: v.size] = v
        v = full
    return v


def quat_mul(a, b):
    return np.array([
        a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
        a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
        a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1],
        a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]])


def quat_to_mat(q):
    w, x, y, z = q
    return np.array([
        [w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z]])


def mat_to_quat(m):
    tr = m[0, 0] + m[1, 1] + m[2, 2]
    if tr > 0:
        s = math.sqrt(tr + 1.0) * 2
        q = np.array([0.25 * s, (m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s, (m[1, 0] - m[0, 1]) / s])
    elif m[0, 0] > m[1, 1] and m[0, 0] > m[2, 2]:
        s = math.sqrt(1.0 + m[0, 0] - m[1, 1] - m[2, 2]) * 2
        q = np.array([(m[2, 1] - m[1, 2]) / s, 0.25 * s, (m[0, 1] + m[1, 0]) / s, (m[0, 2] + m[2, 0]) / s])
    elif m[1, 1] > m[2, 2]:
        s = math.sqrt(1.0 + m[1, 1] - m[0, 0] - m[2, 2]) * 2
        q = np.array([(m[0, 2] - m[2, 0]) / s, (m[0, 1] + m[1, 0]) / s, 0.25 * s, (m[1, 2] + m[2, 1]) / s])
    else:
        s = math.sqrt(1.0 + m[2, 2] - m[0, 0] - m[1, 1]) * 2
        q = np.array([(m[1, 0] - m[0, 1]) / s, (m[0, 2] + m[2, 0]) / s, (m[1, 2] + m[2, 1]) / s, 0.25 * s])
    return q / np.linalg.norm(q)


def axis_angle_quat(axis, ang):
    axis = np.asarray(axis, dtype=np.float64)
    n = np.linalg.norm(axis)
    if n < MINVAL:
        return np.array([1.0, 0, 0, 0])
    s = math.sin(ang / 2) / n
    return np.array([math.cos(ang / 2), axis[0] * s, axis[1] * s, axis[2] * s])


def z_to_quat(vec):
    """Quaternion turning the z axis onto ``vec`` (unit)."""
    axis = np.cross([0.0, 0.0, 1.0], vec)
    s = np.linalg.norm(axis)
    if s < 1e-10:
        axis = np.array([1.0, 0.0, 0.0])
    else:
        axis = axis / s
    ang = math.atan2(s, vec[2])
    return np.array([math.cos(ang / 2), *(axis * math.sin(ang / 2))])


# ----------------------------------------------------------------------------- model container
@dataclass
class Model:
    # sizes
    nq: int = 0
    nv: int = 0
    nu: int = 0
    nbody: int = 0
    njnt: int = 0
    ngeom: int = 0
    nsite: int = 0
    ncam: int = 0
    nsensor: int = 0
    nsensordata: int = 0
    npair: int = 0
    nM: int = 0
    ntree: int = 0
    ndesc: int = 0
    nchild: int = 0
    maxdofdepth: int = 0
    maxtreedof: int = 0
    rowmap: int = 0
    nfactor: int = 0
    npass: int = 0
    ntab: int = 0
    nchunk: int = 0         # chunks of 64 entries of the candidate-pair list
    ntp: int = 0            # pairs of kinematic trees whose bounding-sphere pairs form a block of their own
    nchunk_plane: int = 0   # the list starts with this many chunks of plane pairs, then nchunk_box chunks of (sphere |
    nchunk_box: int = 0     # capsule)-box pairs, then nchunk_boxbox chunks of box-box pairs; the rest are bounding-sphere
    nchunk_boxbox: int = 0  # chunks (sizes, so a specialised kernel knows every chunk's test at compile time)
    # options
    timestep: float = 0.002
    gravity: np.ndarray = field(default_factory=lambda: np.array([0.0, 0.0, -9.81]))
    integrator: int = INT_EULER
    iterations: int = 100
    tolerance: float = 1e-8
    impratio: float = 1.0
    meaninertia: float = 1.0
    nconmax: int = 24        # contact cap per env copy (MuJoCo's <size nconmax>)
    njmax: int = 112         # constraint-row cap per env copy (MuJoCo's <size njmax>)
    # everything else is set by the compiler as numpy arrays
    arrays: dict = field(default_factory=dict)
    names: dict = field(default_factory=dict)
    xml_path: str = ""

    def __getattr__(self, key):
        arrays = self.__dict__.get("arrays", {})
        if key in arrays:
            return arrays[key]
        raise AttributeError(key)

    def name2id(self, kind: str, name: str) -> int:
        try:
            return self.names[kind].index(name)
        except ValueError:
            raise KeyError(f"Invalid name '{name}'. No {kind} with that name.")


# ----------------------------------------------------------------------------- defaults handling
class _Defaults:
    """``<default>`` tree: class name -> {tag: attrib dict}; child classes inherit."""

    def __init__(self, root: ET.Element):
        self.classes = {"main": {}}
        top = root.find("default")
        if top is not None:
            self._walk(top, "main", {})

    def _walk(self, elem, cls_name, inherited):
        own = {tag: dict(attrs) for tag, attrs in inherited.items()}
        for child in elem:
            if child.tag == "default":
                continue
            own.setdefault(child.tag, {}).update(child.attrib)
        self.classes[cls_name] = own
        for child in elem:
            if child.tag == "default":
                self._walk(child, child.attrib.get("class", "main"), own)

    def apply(self, elem: ET.Element, childclass):
        cls = elem.attrib.get("class", childclass or "main")
        merged = dict(self.classes.get(cls, {}).get(elem.tag, {}))
        merged.update(elem.attrib)
        return merged


# ----------------------------------------------------------------------------- the subset, stated
class UnsupportedMJCF(ValueError):
    """The level uses an MJCF feature this compiler does not implement.  ``mujoco.MjModel.from_xml_path``
    (mujoco_parent.py:126) would simulate it; simulating it WITHOUT the feature would be a silently different level,
    so it is refused with the element / attribute named."""


_ORIENT = {"quat", "euler", "axisangle", "xyaxes", "zaxis"}
# top-level sections: compiled, or read by nothing in the simulation (custom: user data; size: allocation hints,
# the caps are config keys here; visual / statistic: rendering scale only)
_SECTIONS = {"compiler", "option", "size", "default", "asset", "visual", "statistic", "worldbody", "actuator", "sensor",
             "custom", "contact", "keyframe"}
# (keyframe: mj_resetData -- the reference's reset, mujoco_parent.py:349 -- returns to qpos0 whatever keys the model holds,
# so the section changes nothing this path computes; contact: <exclude> only, see _pair filter)
# attribute -> accepted values (None: any value is implemented)
_COMPILER_ATTRS = {"angle": {"degree", "radian"}, "coordinate": {"local"}, "inertiafromgeom": {"true", "auto", "false"},
                   "eulerseq": None, "meshdir": None, "texturedir": None, "assetdir": None, "strippath": None,
                   "autolimits": {"false"}, "fusestatic": {"false"}, "discardvisual": {"false"},
                   "balanceinertia": {"false"}, "usethread": None}
_OPTION_ATTRS = {"timestep": None, "gravity": None, "integrator": {"Euler", "RK4"}, "iterations": None,
                 "tolerance": None, "impratio": None, "solver": {"PGS"}, "cone": {"pyramidal"},
                 "jacobian": {"dense", "sparse", "auto"}, "collision": {"all"}, "noslip_iterations": {"0"},
                 "mpr_iterations": None, "mpr_tolerance": None, "magnetic": None}
# option attributes that are accepted only at the value that switches the feature off
_OPTION_ZERO = {"wind", "density", "viscosity", "o_margin"}
_ELEMENT_ATTRS = {
    "body": {"name", "pos", "childclass"} | _ORIENT,
    "inertial": {"pos", "mass", "diaginertia", "fullinertia"} | _ORIENT,
    "geom": {"name", "class", "type", "size", "pos", "fromto", "friction", "density", "mass", "margin", "gap", "condim",
             "contype", "conaffinity", "solref", "solimp", "solmix", "rgba", "material", "group"} | _ORIENT,
    "joint": {"name", "class", "type", "pos", "axis", "limited", "range", "margin", "armature", "damping", "ref",
              "stiffness", "springref", "solreflimit", "solimplimit", "group"},
    "freejoint": {"name", "group"},
    "site": {"name", "class", "pos", "size", "type", "rgba", "group", "material"} | _ORIENT,
    "camera": {"name", "class", "pos", "fovy", "mode"} | _ORIENT,
    "light": {"name", "class", "pos", "dir", "directional", "active", "attenuation", "cutoff", "exponent", "ambient",
              "diffuse", "specular", "castshadow", "mode"},
    "motor": {"name", "class", "joint", "gear", "ctrllimited", "ctrlrange", "group"},
    "general": {"name", "class", "joint", "gear", "ctrllimited", "ctrlrange", "group"},
    "sensor": {"name", "site", "objname", "objtype", "cutoff", "noise", "user"},
}
# attributes that may be present as long as they leave the feature off
_NEUTRAL = {("joint", "frictionloss"): 0.0,
            ("geom", "priority"): 0.0, ("body", "gravcomp"): 0.0, ("body", "mocap"): "false",
            ("motor", "forcelimited"): "false", ("general", "forcelimited"): "false"}
_BODY_CHILDREN = {"body", "geom", "joint", "freejoint", "site", "camera", "light", "inertial"}
_DEFAULT_CHILDREN = {"default", "geom", "joint", "site", "camera", "light", "motor", "general", "material"}
_ASSET_CHILDREN = {"texture", "material"}        # textures are parsed and ignored (rendering: DESIGN 4.2)


def _refuse(what):
    raise UnsupportedMJCF(f"{what} is outside the supported MJCF subset (mjcf.py); the level would be simulated "
                          "without it, so it is refused")


def _check_attrs(el, tag=None):
    tag = tag or el.tag
    allowed = _ELEMENT_ATTRS[tag]
    for key, value in el.attrib.items():
        if key in allowed:
            continue
        off = _NEUTRAL.get((tag, key))
        if off is not None:
            try:
                same = (value == off) if isinstance(off, str) else all(float(t) == off for t in value.split())
            except ValueError:
                same = False
            if same:
                continue
        _refuse(f'attribute {key}="{value}" of <{el.tag}>')


def check_subset(root: ET.Element):
    """Raise :class:`UnsupportedMJCF` for anything in the document this compiler would otherwise skip over."""
    if root.tag != "mujoco":
        _refuse(f"root element <{root.tag}>")
    for section in root:
        if section.tag not in _SECTIONS:
            _refuse(f"section <{section.tag}>")
    for comp in root.findall("compiler"):
        for key, value in comp.attrib.items():
            if key not in _COMPILER_ATTRS:
                _refuse(f'<compiler {key}="{value}">')
            if _COMPILER_ATTRS[key] is not None and value not in _COMPILER_ATTRS[key]:
                _refuse(f'<compiler {key}="{value}">')
    for opt in root.findall("option"):
        for key, value in opt.attrib.items():
            if key in _OPTION_ZERO:
                if any(float(t) != 0.0 for t in value.split()):
                    _refuse(f'<option {key}="{value}">')
                continue
            if key not in _OPTION_ATTRS:
                _refuse(f'<option {key}="{value}">')
            if _OPTION_ATTRS[key] is not None and value not in _OPTION_ATTRS[key]:
                _refuse(f'<option {key}="{value}">' + (" (this stepper solves with PGS on the pyramidal cone, "
                                                        "BASELINE.json north_star)" if key in ("solver", "cone") else ""))
        for child in opt:
            # <flag>: every flag switches a stage of the step on or off
            if child.tag == "flag" and all(v == ("disable" if k in _FLAGS_OFF else "enable")
                                           for k, v in child.attrib.items()):
                continue
            _refuse(f"<option><{child.tag} {' '.join(child.attrib)}>")

    def defaults(elem):
        for child in elem:
            if child.tag not in _DEFAULT_CHILDREN:
                _refuse(f"<default><{child.tag}>")
            if child.tag == "default":
                defaults(child)
            elif child.tag != "material":
                _check_attrs(child)
    for top in root.findall("default"):
        defaults(top)
    for asset in root.findall("asset"):
        for child in asset:
            if child.tag not in _ASSET_CHILDREN:
                _refuse(f"<asset><{child.tag}>")

    def body(elem):
        for child in elem:
            if child.tag not in _BODY_CHILDREN:
                _refuse(f"<{child.tag}> inside <{elem.tag}>")
            _check_attrs(child)
            if child.tag == "body":
                body(child)
    for world in root.findall("worldbody"):
        body(world)
    for con in root.findall("contact"):
        for el in con:
            # (explicit <pair>s carry their own solver parameters and dimensions: not implemented)
            if el.tag != "exclude" or set(el.attrib) - {"name", "body1", "body2"}:
                _refuse(f"<contact><{el.tag} {' '.join(el.attrib)}>")
    for act in root.findall("actuator"):
        for el in act:
            if el.tag not in ("motor", "general"):
                _refuse(f"actuator <{el.tag}>")
            _check_attrs(el)
    for sens in root.findall("sensor"):
        for el in sens:
            if el.tag not in SENSOR_TYPES:
                _refuse(f"sensor <{el.tag}>")
            _check_attrs(el, "sensor")


# <option><flag>: MuJoCo's disable flags default to "enable", its enable flags (these) to "disable"
_FLAGS_OFF = {"override", "energy", "fwdinv", "sensornoise", "multiccd", "island"}


# ----------------------------------------------------------------------------- compiler
class _Compiler:
    def __init__(self, xml_text: str):
        self.root = ET.fromstring(xml_text)
        check_subset(self.root)
        comp = self.root.find("compiler")
        comp = comp.attrib if comp is not None else {}
        # inertiafromgeom: "true" infer every body's inertia from its geoms, "false" use <inertial> only, "auto" (the
        # default) infer where a body has no <inertial>
        self.inertiafromgeom = comp.get("inertiafromgeom", "auto")
        self.degree = comp.get("angle", "degree") == "degree"
        self.eulerseq = comp.get("eulerseq", "xyz")
        self.defaults = _Defaults(self.root)
        self.bodies, self.joints, self.geoms, self.sites, self.cams = [], [], [], [], []
        self.lights = []
        # <asset><material>: the part of a material the fixed-function shading uses (colour, specular, shininess,
        # emission; XML reference, asset/material).  Textures and reflectance are parsed and ignored.
        self.materials = {}
        asset = self.root.find("asset")
        if asset is not None:
            for el in asset:
                if el.tag == "material":
                    a = self.defaults.apply(el, None)
                    self.materials[a.get("name", "")] = dict(
                        rgba=_vec(a.get("rgba"), 4, [1.0, 1.0, 1.0, 1.0]), specular=float(a.get("specular", 0.5)),
                        shininess=float(a.get("shininess", 0.5)), emission=float(a.get("emission", 0.0)))
        # <visual><headlight>: XML reference defaults ambient 0.1, diffuse 0.4, specular 0.5, active 1
        self.headlight = dict(active=1.0, ambient=np.full(3, 0.1), diffuse=np.full(3, 0.4), specular=np.full(3, 0.5))
        visual = self.root.find("visual")
        if visual is not None and visual.find("headlight") is not None:
            a = visual.find("headlight").attrib
            self.headlight = dict(active=float(a.get("active", 1)), ambient=_vec(a.get("ambient"), 3, [0.1] * 3),
                                  diffuse=_vec(a.get("diffuse"), 3, [0.4] * 3), specular=_vec(a.get("specular"), 3, [0.5] * 3))

    # -- orientation attributes -------------------------------------------------
    def _angle(self, a):
        return math.radians(a) if self.degree else a

    def orientation(self, attrs):
        if "quat" in attrs:
            q = _vec(attrs["quat"])
            return q / np.linalg.norm(q)
        if "euler" in attrs:
            e = _vec(attrs["euler"])
            q = np.array([1.0, 0, 0, 0])
            for ax_char, ang in zip(self.eulerseq, e):
                axis = np.zeros(3)
                axis["xyz".index(ax_char.lower())] = 1.0
                rot = axis_angle_quat(axis, self._angle(ang))
                q = quat_mul(q, rot) if ax_char.islower() else quat_mul(rot, q)
            return q / np.linalg.norm(q)
        if "axisangle" in attrs:
            aa = _vec(attrs["axisangle"])
            return axis_angle_quat(aa[:3], self._angle(aa[3]))
        if "xyaxes" in attrs:
            xy = _vec(attrs["xyaxes"])
            x = xy[:3] / np.linalg.norm(xy[:3])
            y = xy[3:] - np.dot(xy[3:], x) * x
            y /= np.linalg.norm(y)
            z = np.cross(x, y)
            return mat_to_quat(np.stack([x, y, z], axis=1))
        if "zaxis" in attrs:
            z = _vec(attrs["zaxis"])
            return z_to_quat(z / np.linalg.norm(z))
        return np.array([1.0, 0, 0, 0])

    # -- tree walk ---------------------------------------------------------------
    def walk(self):
        world = self.root.find("worldbody")
        self.bodies.append(dict(name="world", parent=0, pos=np.zeros(3), quat=np.array([1.0, 0, 0, 0]),
                                jnts=[], geoms=[]))
        self._body_children(world, 0, None)

    def _body_children(self, elem, body_id, childclass):
        # MuJoCo numbers bodies depth-first in document order; joints, geoms, sites and cameras are
        # numbered in the order their owning bodies are numbered.
        for child in elem:
            if child.tag == "joint" or child.tag == "freejoint":
                self._add_joint(child, body_id, childclass)
            elif child.tag == "geom":
                self._add_geom(child, body_id, childclass)
            elif child.tag == "site":
                a = self.defaults.apply(child, childclass)
                self.sites.append(dict(name=a.get("name", ""), body=body_id,
                                       pos=_vec(a.get("pos"), 3, [0, 0, 0]), quat=self.orientation(a),
                                       size=_vec(a.get("size"), 3, [0.005, 0.005, 0.005]),
                                       type=a.get("type", "sphere")))
            elif child.tag == "inertial":
                # XML reference, body/inertial: pos and mass required; diaginertia in the frame given by the orientation
                # attributes, or fullinertia (M11 M22 M33 M12 M13 M23) in the body frame
                a = child.attrib
                if "pos" not in a or "mass" not in a or not ("diaginertia" in a or "fullinertia" in a):
                    raise ValueError("<inertial> needs pos, mass and diaginertia or fullinertia")
                self.bodies[body_id]["inertial"] = dict(
                    pos=_vec(a["pos"]), quat=self.orientation(a), mass=float(a["mass"]),
                    diag=_vec(a["diaginertia"]) if "diaginertia" in a else None,
                    full=_vec(a["fullinertia"]) if "fullinertia" in a else None)
            elif child.tag == "camera":
                a = self.defaults.apply(child, childclass)
                self.cams.append(dict(name=a.get("name", ""), body=body_id,
                                      pos=_vec(a.get("pos"), 3, [0, 0, 0]), quat=self.orientation(a),
                                      fovy=float(a.get("fovy", 45.0)), mode=a.get("mode", "fixed")))
            elif child.tag == "light":
                # XML reference, body/light: directional false, active true, pos 0 0 0, dir 0 0 -1, attenuation 1 0 0,
                # cutoff 45, exponent 10, ambient 0 0 0, diffuse 0.7 0.7 0.7, specular 0.3 0.3 0.3, castshadow true
                a = self.defaults.apply(child, childclass)
                if a.get("active", "true") == "false":
                    continue
                if a.get("mode", "fixed") != "fixed":       # (track / trackcom / targetbody...: the light would move or turn)
                    _refuse(f'light mode "{a.get("mode")}"')
                d = _vec(a.get("dir"), 3, [0, 0, -1])
                n = np.linalg.norm(d)
                self.lights.append(dict(name=a.get("name", ""), body=body_id, pos=_vec(a.get("pos"), 3, [0, 0, 0]),
                                        dir=d / n if n > MINVAL else np.array([0.0, 0, -1]),
                                        directional=int(a.get("directional", "false") == "true"),
                                        castshadow=int(a.get("castshadow", "true") == "true"),
                                        attenuation=_vec(a.get("attenuation"), 3, [1, 0, 0]),
                                        cutoff=float(a.get("cutoff", 45.0)), exponent=float(a.get("exponent", 10.0)),
                                        ambient=_vec(a.get("ambient"), 3, [0, 0, 0]),
                                        diffuse=_vec(a.get("diffuse"), 3, [0.7, 0.7, 0.7]),
                                        specular=_vec(a.get("specular"), 3, [0.3, 0.3, 0.3])))
        for child in elem:
            if child.tag == "body":
                a = child.attrib
                new_id = len(self.bodies)
                self.bodies.append(dict(name=a.get("name", ""), parent=body_id,
                                        pos=_vec(a.get("pos"), 3, [0, 0, 0]), quat=self.orientation(a),
                                        jnts=[], geoms=[]))
                self._body_children(child, new_id, a.get("childclass", childclass))

    def fuse_static(self):
        """Fold every body that cannot move -- no joint on it or on any ancestor -- into the world body: its geoms, sites,
        cameras and lights become the world's, at their composed poses.  What MuJoCo's ``<compiler fusestatic>`` does; here
        it is applied only when a level has more bodies than a wavefront has lanes (the arena's walls and boxes are one
        body each).  Physics is unchanged -- a static body has no dofs, and contacts against it are contacts against the
        world --, geom ids are unchanged, body ids shrink.  Returns {name: record} of the folded bodies so that host
        queries by body name (``get_data``, ``distance``, tags) still answer: position = the body's inertial position."""
        nb = len(self.bodies)
        static = [False] * nb
        static[0] = True
        for b in range(1, nb):
            static[b] = static[self.bodies[b]["parent"]] and not self.bodies[b]["jnts"]
        # world pose of every static body (parents come first)
        wpos, wquat = [np.zeros(3)] * nb, [np.array([1.0, 0, 0, 0])] * nb
        for b in range(1, nb):
            if static[b]:
                p = self.bodies[b]["parent"]
                wpos[b] = wpos[p] + quat_to_mat(wquat[p]) @ self.bodies[b]["pos"]
                wquat[b] = quat_mul(wquat[p], self.bodies[b]["quat"])
        folded = {}
        for b in range(1, nb):
            if not static[b]:
                continue
            body = self.bodies[b]
            rot = quat_to_mat(wquat[b])
            for kind in (self.geoms, self.sites, self.cams, self.lights):
                for item in kind:
                    if item["body"] == b:
                        item["pos"] = wpos[b] + rot @ item["pos"]
                        if "quat" in item:
                            item["quat"] = quat_mul(wquat[b], item["quat"])
                        if "dir" in item:
                            item["dir"] = rot @ item["dir"]
                        item["body"] = 0
            self.bodies[0]["geoms"] += body["geoms"]
            if body["name"]:
                # (inertial position: the mass-weighted centre of its geoms, now in world coordinates)
                parts = [(_geom_mass_inertia(self.geoms[g])[0], self.geoms[g]["pos"]) for g in body["geoms"]]
                total = sum(mass for mass, _ in parts)
                xipos = sum(mass * pos for mass, pos in parts) / total if total > 0 else wpos[b].copy()
                folded[body["name"]] = dict(pos=wpos[b].copy(), quat=wquat[b].copy(), geoms=list(body["geoms"]),
                                            xipos=np.asarray(xipos, np.float64), mass=float(total))
        self.bodies[0]["geoms"].sort()
        # moving bodies whose parent was folded hang off the world at their composed pose
        for b in range(1, nb):
            p = self.bodies[b]["parent"]
            if not static[b] and p != 0 and static[p]:
                self.bodies[b]["pos"] = wpos[p] + quat_to_mat(wquat[p]) @ self.bodies[b]["pos"]
                self.bodies[b]["quat"] = quat_mul(wquat[p], self.bodies[b]["quat"])
                self.bodies[b]["parent"] = 0
        # renumber
        new_id, kept = {}, []
        for b in range(nb):
            if b == 0 or not static[b]:
                new_id[b] = len(kept)
                kept.append(self.bodies[b])
        for body in kept:
            body["parent"] = new_id[body["parent"]]
        for kind in (self.geoms, self.sites, self.cams, self.lights, self.joints):
            for item in kind:
                item["body"] = new_id[item["body"]]
        self.bodies = kept
        return folded

    def _add_joint(self, elem, body_id, childclass):
        if elem.tag == "freejoint":
            a = dict(elem.attrib)
            a["type"] = "free"
            a.setdefault("limited", "false")
            a.setdefault("armature", "0")
            a.setdefault("damping", "0")
        else:
            a = self.defaults.apply(elem, childclass)
        if a.get("type", "hinge") not in JNT_TYPES:
            _refuse(f'joint type "{a.get("type")}"')
        jtype = JNT_TYPES[a.get("type", "hinge")]
        limited = a.get("limited", "false") == "true"
        rng = _vec(a.get("range"), 2, [0, 0])
        if jtype in (JNT_HINGE,) and self.degree:
            rng = np.radians(rng)
        axis = _vec(a.get("axis"), 3, [0, 0, 1])
        n = np.linalg.norm(axis)
        axis = axis / n if n > MINVAL else np.array([0.0, 0, 1])
        if "solreflimit" in a and min(_vec(a["solreflimit"])) <= 0:
            _refuse(f'joint solreflimit "{a["solreflimit"]}" (the direct stiffness / damping form; positive timeconst and dampratio only)')
        j = dict(name=a.get("name", ""), type=jtype, body=body_id, pos=_vec(a.get("pos"), 3, [0, 0, 0]),
                 axis=axis, limited=limited and jtype != JNT_FREE, range=rng,
                 margin=float(a.get("margin", 0.0)), armature=float(a.get("armature", 0.0)),
                 damping=float(a.get("damping", 0.0)), stiffness=float(a.get("stiffness", 0.0)),
                 ref=float(a.get("ref", 0.0)), springref=float(a.get("springref", 0.0)),
                 solref=_vec(a.get("solreflimit"), 2, [0.02, 1.0]),
                 solimp=_vec(a.get("solimplimit"), 5, [0.9, 0.95, 0.001, 0.5, 2.0]))
        if jtype == JNT_FREE and j["stiffness"] != 0.0:
            _refuse("stiffness of a free joint")
        if jtype == JNT_HINGE and self.degree:
            j["springref"] = math.radians(j["springref"])         # (an angle like ref and range: the compiler's unit)
            j["ref"] = math.radians(j["ref"])
        self.bodies[body_id]["jnts"].append(len(self.joints))
        self.joints.append(j)

    def _add_geom(self, elem, body_id, childclass):
        a = self.defaults.apply(elem, childclass)
        if a.get("type", "sphere") not in GEOM_TYPES:
            _refuse(f'geom type "{a.get("type")}"')
        gtype = GEOM_TYPES[a.get("type", "sphere")]
        size = _vec(a.get("size"), 3, [0, 0, 0])
        pos = _vec(a.get("pos"), 3, [0, 0, 0])
        quat = self.orientation(a)
        if "fromto" in a:
            ft = _vec(a["fromto"])
            vec = ft[:3] - ft[3:]
            length = np.linalg.norm(vec)
            pos = 0.5 * (ft[:3] + ft[3:])
            quat = z_to_quat(vec / length)
            size = np.array([size[0], 0.5 * length, 0.0])
        # (solref <= 0 is MuJoCo's direct form, (-stiffness, -damping); the row builder implements (timeconst, dampratio))
        if "solref" in a and min(_vec(a["solref"])) <= 0:
            _refuse(f'geom solref "{a["solref"]}" (the direct stiffness / damping form; positive timeconst and dampratio only)')
        # (condim 4 / 6 add torsional / rolling friction rows; the row builder makes the normal row or the 4-edge pyramid)
        if int(a.get("condim", 3)) not in (1, 3):
            _refuse(f'geom condim {a.get("condim")} (1 and 3 only: torsional and rolling friction are not implemented)')
        g = dict(name=a.get("name", ""), type=gtype, body=body_id, size=size, pos=pos, quat=quat,
                 friction=_vec(a.get("friction"), 3, [1.0, 0.005, 0.0001]),
                 density=float(a.get("density", 1000.0)), margin=float(a.get("margin", 0.0)),
                 gap=float(a.get("gap", 0.0)), condim=int(a.get("condim", 3)),
                 contype=int(a.get("contype", 1)), conaffinity=int(a.get("conaffinity", 1)),
                 solref=_vec(a.get("solref"), 2, [0.02, 1.0]),
                 solimp=_vec(a.get("solimp"), 5, [0.9, 0.95, 0.001, 0.5, 2.0]),
                 solmix=float(a.get("solmix", 1.0)),
                 rgba=_vec(a.get("rgba"), 4, [0.5, 0.5, 0.5, 1.0]),
                 mass=float(a["mass"]) if "mass" in a else None, group=int(a.get("group", 0)))
        # visual material (XML reference, geom/material and geom/rgba): without a material the renderer's defaults
        # (specular 0.5, shininess 0.5, no emission); with one, its properties -- and its colour unless the geom sets rgba
        mat = self.materials.get(a.get("material", ""))
        g["matprop"] = np.array([mat["specular"], mat["shininess"], mat["emission"]] if mat else [0.5, 0.5, 0.0])
        if mat and "rgba" not in a:
            g["rgba"] = mat["rgba"].copy()
        self.bodies[body_id]["geoms"].append(len(self.geoms))
        self.geoms.append(g)


def _geom_mass_inertia(g):
    """Mass and principal inertia (geom frame) of a solid primitive of uniform density."""
    t, s, rho = g["type"], g["size"], g["density"]
    if t == GEOM_SPHERE:
        r = s[0]
        vol = 4.0 / 3.0 * math.pi * r ** 3
        unit = np.array([0.4 * r * r] * 3)
        mass = rho * vol
        inertia = mass * unit
    elif t == GEOM_CAPSULE:
        r, h = s[0], 2.0 * s[1]
        m_cyl = rho * math.pi * r * r * h
        m_sph = rho * 4.0 / 3.0 * math.pi * r ** 3
        mass = m_cyl + m_sph
        lateral = m_cyl * (3 * r * r + h * h) / 12.0 + m_sph * (0.4 * r * r + 0.25 * h * h + 0.375 * h * r)
        axial = m_cyl * r * r / 2.0 + m_sph * 0.4 * r * r
        inertia = np.array([lateral, lateral, axial])
    elif t == GEOM_BOX:
        mass = rho * 8.0 * s[0] * s[1] * s[2]
        inertia = mass / 3.0 * np.array([s[1] ** 2 + s[2] ** 2, s[0] ** 2 + s[2] ** 2, s[0] ** 2 + s[1] ** 2])
    else:  # plane: no volume
        return 0.0, np.zeros(3)
    if g["mass"] is not None and mass > 0:
        inertia = inertia * (g["mass"] / mass)
        mass = g["mass"]
    return mass, inertia


def body_subtree_massless(b, parentid, mass):
    """No body of b's subtree has mass (a moving body may be massless itself when it carries something that is not)."""
    for k in range(b, len(parentid)):
        p = k
        while p > b:
            p = int(parentid[p])
        if p == b and mass[k] >= MINVAL:
            return False
    return True


def _geom_rbound(g):
    t, s = g["type"], g["size"]
    if t == GEOM_SPHERE:
        return s[0]
    if t == GEOM_CAPSULE:
        return s[0] + s[1]
    if t == GEOM_BOX:
        return float(np.linalg.norm(s))
    return 0.0


SUPPORTED_PAIRS = {
    (GEOM_PLANE, GEOM_SPHERE), (GEOM_PLANE, GEOM_CAPSULE), (GEOM_PLANE, GEOM_BOX),
    (GEOM_SPHERE, GEOM_SPHERE), (GEOM_SPHERE, GEOM_CAPSULE), (GEOM_SPHERE, GEOM_BOX),
    (GEOM_CAPSULE, GEOM_CAPSULE), (GEOM_CAPSULE, GEOM_BOX), (GEOM_BOX, GEOM_BOX),
}


MAX_LANE_BODIES = 64      # a wavefront's lanes: bodies, joints and dofs of a level must fit (geoms: two passes, 128)


def compile_mjcf(xml_path: str, nconmax: int | None = None, njmax: int | None = None, lane_map: bool = True,
                 broad_cull: bool = True, fuse_static: bool | None = None) -> Model:
    with open(xml_path, "r") as fh:
        text = fh.read()
    return compile_mjcf_string(text, xml_path=xml_path, nconmax=nconmax, njmax=njmax, lane_map=lane_map, broad_cull=broad_cull,
                               fuse_static=fuse_static)


def compile_mjcf_string(text: str, xml_path: str = "", nconmax=None, njmax=None, lane_map: bool = True,
                        broad_cull: bool = True, fuse_static: bool | None = None) -> Model:
    """``lane_map=False`` withholds the tree-row lane map even from a model that qualifies for it (a model with more
    than four trees or more than 16 dofs in a tree never gets it): the kernels then take their general paths.
    ``fuse_static``: fold the bodies that cannot move into the world (``_Compiler.fuse_static``); ``None`` = only when
    the level has more than 64 bodies, so that every level that fits keeps MuJoCo's body ids."""
    c = _Compiler(text)
    c.walk()
    folded = {}
    if fuse_static or (fuse_static is None and len(c.bodies) > MAX_LANE_BODIES):
        folded = c.fuse_static()
    m = Model(xml_path=xml_path)
    m.folded_bodies = folded
    A = m.arrays
    opt = c.root.find("option")
    opt = opt.attrib if opt is not None else {}
    m.timestep = float(opt.get("timestep", 0.002))
    m.gravity = _vec(opt.get("gravity"), 3, [0, 0, -9.81])
    m.integrator = INT_RK4 if opt.get("integrator", "Euler") == "RK4" else INT_EULER
    m.iterations = int(opt.get("iterations", 100))
    m.tolerance = float(opt.get("tolerance", 1e-8))
    m.impratio = float(opt.get("impratio", 1.0))

    nbody, njnt, ngeom = len(c.bodies), len(c.joints), len(c.geoms)
    m.nbody, m.njnt, m.ngeom, m.nsite, m.ncam = nbody, njnt, ngeom, len(c.sites), len(c.cams)
    # (geom groups 3..5 are hidden under the default visual options the reference renders with, mujoco_parent.py:533; the
    # ray caster draws every opaque geom -- a level that relies on hidden groups would get other camera images)
    if c.cams and any(g["group"] >= 3 for g in c.geoms):
        _refuse("a geom in group 3 or higher (hidden by the renderer's default options) in a level with cameras")

    # ---- joints / dofs
    jnt_qposadr, jnt_dofadr = np.zeros(njnt, np.int32), np.zeros(njnt, np.int32)
    nq = nv = 0
    for j, jn in enumerate(c.joints):
        jnt_qposadr[j], jnt_dofadr[j] = nq, nv
        nq += 7 if jn["type"] == JNT_FREE else 1
        nv += 6 if jn["type"] == JNT_FREE else 1
    m.nq, m.nv = nq, nv
    A["jnt_type"] = np.array([j["type"] for j in c.joints], np.int32)
    A["jnt_qposadr"], A["jnt_dofadr"] = jnt_qposadr, jnt_dofadr
    A["jnt_bodyid"] = np.array([j["body"] for j in c.joints], np.int32)
    A["jnt_pos"] = np.array([j["pos"] for j in c.joints], np.float64).reshape(njnt, 3)
    A["jnt_axis"] = np.array([j["axis"] for j in c.joints], np.float64).reshape(njnt, 3)
    A["jnt_limited"] = np.array([int(j["limited"]) for j in c.joints], np.int32)
    A["jnt_range"] = np.array([j["range"] for j in c.joints], np.float64).reshape(njnt, 2)
    A["jnt_margin"] = np.array([j["margin"] for j in c.joints], np.float64)
    A["jnt_solref"] = np.array([j["solref"] for j in c.joints], np.float64).reshape(njnt, 2)
    A["jnt_solimp"] = np.array([j["solimp"] for j in c.joints], np.float64).reshape(njnt, 5)

    qpos0 = np.zeros(nq)
    dof_bodyid, dof_jntid = np.zeros(nv, np.int32), np.zeros(nv, np.int32)
    dof_armature, dof_damping = np.zeros(nv), np.zeros(nv)
    # joint springs (XML reference, body/joint stiffness and springref): the passive force -stiffness (q - springref) on a
    # hinge's or a slide's coordinate
    dof_stiffness, dof_springref, dof_qposadr = np.zeros(nv), np.zeros(nv), np.zeros(nv, np.int32)

    # ---- bodies
    body_parentid = np.array([b["parent"] for b in c.bodies], np.int32)
    body_jntnum = np.array([len(b["jnts"]) for b in c.bodies], np.int32)
    body_jntadr = np.array([b["jnts"][0] if b["jnts"] else -1 for b in c.bodies], np.int32)
    body_geomnum = np.array([len(b["geoms"]) for b in c.bodies], np.int32)
    body_geomadr = np.array([b["geoms"][0] if b["geoms"] else -1 for b in c.bodies], np.int32)
    body_dofnum, body_dofadr = np.zeros(nbody, np.int32), np.full(nbody, -1, np.int32)
    body_pos = np.array([b["pos"] for b in c.bodies], np.float64).reshape(nbody, 3)
    body_quat = np.array([b["quat"] for b in c.bodies], np.float64).reshape(nbody, 4)
    body_rootid, body_weldid = np.zeros(nbody, np.int32), np.zeros(nbody, np.int32)
    body_depth = np.zeros(nbody, np.int32)
    body_lastdof = np.full(nbody, -1, np.int32)   # last dof on the path from the world to this body
    dof_parentid = np.full(nv, -1, np.int32)
    for b in range(1, nbody):
        p = body_parentid[b]
        body_rootid[b] = b if p == 0 else body_rootid[p]
        body_depth[b] = body_depth[p] + 1
        body_weldid[b] = b if body_jntnum[b] > 0 else body_weldid[p]
        last = body_lastdof[p]
        for j in c.bodies[b]["jnts"]:
            jn = c.joints[j]
            nd = 6 if jn["type"] == JNT_FREE else 1
            if body_dofadr[b] < 0:
                body_dofadr[b] = jnt_dofadr[j]
            body_dofnum[b] += nd
            for k in range(nd):
                d = jnt_dofadr[j] + k
                dof_bodyid[d], dof_jntid[d] = b, j
                dof_parentid[d] = last
                dof_armature[d], dof_damping[d] = jn["armature"], jn["damping"]
                dof_qposadr[d] = jnt_qposadr[j] + (k if jn["type"] != JNT_FREE else min(k, 2))
                if jn["type"] != JNT_FREE:
                    dof_stiffness[d], dof_springref[d] = jn["stiffness"], jn["springref"]
                last = d
            if jn["type"] == JNT_FREE:
                qa = jnt_qposadr[j]
                # a free joint's reference pose is the body's frame in its parent (the parent is the world)
                qpos0[qa:qa + 3] = body_pos[b]
                qpos0[qa + 3:qa + 7] = body_quat[b]
            else:
                qpos0[jnt_qposadr[j]] = jn["ref"]
        body_lastdof[b] = last
    # Kinematic parents: a body whose parent carries no joint is welded to it, so its frame (and its spatial velocity)
    # follows from the nearest ancestor that does move, through a constant offset composed here.  The kernels walk the
    # tree by these links: fewer levels (the ant: torso - leg - hip - ankle becomes torso - {leg, hip} - ankle).
    body_kparent, body_kdepth = body_parentid.copy(), np.zeros(nbody, np.int32)
    body_kpos, body_kquat = body_pos.copy(), body_quat.copy()
    for b in range(1, nbody):
        p, pos, quat = int(body_parentid[b]), body_pos[b].copy(), body_quat[b].copy()
        while p != 0 and body_jntnum[p] == 0:
            pos = body_pos[p] + quat_to_mat(body_quat[p]) @ pos
            quat = quat_mul(body_quat[p], quat)
            p = int(body_parentid[p])
        body_kparent[b], body_kpos[b], body_kquat[b] = p, pos, quat
        body_kdepth[b] = body_kdepth[p] + 1
    A.update(body_kparent=body_kparent, body_kdepth=body_kdepth, body_kpos=body_kpos, body_kquat=body_kquat)
    A.update(body_parentid=body_parentid, body_rootid=body_rootid, body_weldid=body_weldid,
             body_jntnum=body_jntnum, body_jntadr=body_jntadr, body_dofnum=body_dofnum,
             body_dofadr=body_dofadr, body_geomnum=body_geomnum, body_geomadr=body_geomadr,
             body_pos=body_pos, body_quat=body_quat, body_depth=body_depth, body_lastdof=body_lastdof,
             dof_bodyid=dof_bodyid, dof_jntid=dof_jntid, dof_parentid=dof_parentid,
             dof_armature=dof_armature, dof_damping=dof_damping, qpos0=qpos0,
             dof_stiffness=dof_stiffness, dof_springref=dof_springref, dof_qposadr=dof_qposadr)

    # kinematic trees: one per child of the world that carries (or whose subtree carries) dofs
    dof_treeid = np.zeros(nv, np.int32)
    roots = sorted({int(body_rootid[dof_bodyid[d]]) for d in range(nv)})
    for d in range(nv):
        dof_treeid[d] = roots.index(int(body_rootid[dof_bodyid[d]]))
    m.ntree = len(roots)
    A["dof_treeid"] = dof_treeid
    body_treeid = np.full(nbody, -1, np.int32)
    for b in range(1, nbody):
        if int(body_rootid[b]) in roots:
            body_treeid[b] = roots.index(int(body_rootid[b]))
    A["body_treeid"] = body_treeid

    # sparse inertia addressing: row d holds the diagonal followed by its ancestors, walking up
    dof_Madr = np.zeros(nv, np.int32)
    dof_depth = np.zeros(nv, np.int32)
    nM = 0
    for d in range(nv):
        dof_Madr[d] = nM
        k, n = d, 0
        while k >= 0:
            n += 1
            k = dof_parentid[k]
        dof_depth[d] = n - 1
        nM += n
    m.nM = nM
    A["dof_Madr"], A["dof_depth"] = dof_Madr, dof_depth
    # per sparse entry: its row and column dof; per dof: the entries L[k][d] of its descendants k (ascending k)
    M_rowid, M_colid = np.zeros(nM, np.int32), np.zeros(nM, np.int32)
    desc = [[] for _ in range(nv)]
    for d in range(nv):
        k, adr = d, dof_Madr[d]
        while k >= 0:
            M_rowid[adr], M_colid[adr] = d, k
            if k != d:
                desc[k].append(adr)
            adr += 1
            k = dof_parentid[k]
    A["M_rowid"], A["M_colid"] = M_rowid, M_colid
    A["dof_descadr"] = np.cumsum([0] + [len(x) for x in desc[:-1]]).astype(np.int32) if nv else np.zeros(0, np.int32)
    A["dof_descnum"] = np.array([len(x) for x in desc], np.int32)
    A["desc_Madr"] = np.array([a for x in desc for a in x], np.int32)
    A["desc_row"] = M_rowid[A["desc_Madr"]] if A["desc_Madr"].size else np.zeros(0, np.int32)
    # 64-bit set per dof: the dof itself and every dof below it (low word, high word); dof a lies on the ancestor
    # chain of dof x exactly when bit x of a's set is on -- the test the compact constraint rows are read with
    mask = np.zeros(nv, np.uint64)
    for d in range(nv):
        k = d
        while k >= 0:
            mask[k] |= np.uint64(1) << np.uint64(d)
            k = dof_parentid[k]
    A["dof_descmask"] = np.stack([(mask & np.uint64(0xFFFFFFFF)).astype(np.uint32),
                                  (mask >> np.uint64(32)).astype(np.uint32)], axis=1).reshape(-1).view(np.int32)
    A["M_coldiag"] = dof_Madr[M_colid] if nM else np.zeros(0, np.int32)   # address of the diagonal of each entry's column
    m.ndesc = int(A["desc_Madr"].size)
    m.maxdofdepth = int(dof_depth.max()) if nv else 0
    # children of every body (descending id, the order in which a backward pass over bodies meets them)
    kids = [[] for _ in range(nbody)]
    for b in range(nbody - 1, 0, -1):
        kids[body_parentid[b]].append(b)
    A["body_childadr"] = np.cumsum([0] + [len(x) for x in kids[:-1]]).astype(np.int32)
    A["body_childnum"] = np.array([len(x) for x in kids], np.int32)
    A["body_childid"] = np.array([c for x in kids for c in x], np.int32)
    m.nchild = int(A["body_childid"].size)
    A["tree_rootbody"] = np.array(roots, np.int32)
    # bodies are numbered depth-first, so a body's subtree is the id range [b, b + body_subtreenum[b])
    subtreenum = np.ones(nbody, np.int32)
    for b in range(nbody - 1, 0, -1):
        subtreenum[body_parentid[b]] += subtreenum[b]
    A["body_subtreenum"] = subtreenum
    # dofs of a tree are contiguous: first dof and dof count per tree
    A["tree_dofadr"] = np.array([int(np.min(np.nonzero(dof_treeid == t)[0])) for t in range(len(roots))], np.int32)
    A["tree_dofnum"] = np.array([int(np.sum(dof_treeid == t)) for t in range(len(roots))], np.int32)
    m.maxtreedof = int(A["tree_dofnum"].max()) if len(roots) else 0

    # ---- geoms
    A["geom_type"] = np.array([g["type"] for g in c.geoms], np.int32)
    A["geom_bodyid"] = np.array([g["body"] for g in c.geoms], np.int32)
    A["geom_size"] = np.array([g["size"] for g in c.geoms], np.float64).reshape(ngeom, 3)
    A["geom_pos"] = np.array([g["pos"] for g in c.geoms], np.float64).reshape(ngeom, 3)
    A["geom_quat"] = np.array([g["quat"] for g in c.geoms], np.float64).reshape(ngeom, 4)
    A["geom_friction"] = np.array([g["friction"] for g in c.geoms], np.float64).reshape(ngeom, 3)
    A["geom_margin"] = np.array([g["margin"] for g in c.geoms], np.float64)
    A["geom_gap"] = np.array([g["gap"] for g in c.geoms], np.float64)
    A["geom_condim"] = np.array([g["condim"] for g in c.geoms], np.int32)
    A["geom_solref"] = np.array([g["solref"] for g in c.geoms], np.float64).reshape(ngeom, 2)
    A["geom_solimp"] = np.array([g["solimp"] for g in c.geoms], np.float64).reshape(ngeom, 5)
    A["geom_solmix"] = np.array([g["solmix"] for g in c.geoms], np.float64)
    A["geom_rbound"] = np.array([_geom_rbound(g) for g in c.geoms], np.float64)
    A["geom_rgba"] = np.array([g["rgba"] for g in c.geoms], np.float64).reshape(ngeom, 4)
    A["geom_matprop"] = np.array([g["matprop"] for g in c.geoms], np.float64).reshape(ngeom, 3)

    # ---- body inertial frames from geoms
    body_mass, body_ipos = np.zeros(nbody), np.zeros((nbody, 3))
    body_iquat, body_inertia = np.tile([1.0, 0, 0, 0], (nbody, 1)), np.zeros((nbody, 3))
    def principal(tensor):
        """Principal moments (descending) and the right-handed frame of a symmetric inertia tensor."""
        evals, evecs = np.linalg.eigh(tensor)
        order = np.argsort(-evals)
        evals, evecs = evals[order], evecs[:, order]
        if np.linalg.det(evecs) < 0:
            evecs[:, 2] = -evecs[:, 2]
        return evals, mat_to_quat(evecs)
    for b in range(1, nbody):
        given = c.bodies[b].get("inertial")
        if given is not None and c.inertiafromgeom != "true":
            body_mass[b], body_ipos[b] = given["mass"], given["pos"]
            if given["full"] is not None:
                f = given["full"]
                body_inertia[b], body_iquat[b] = principal(np.array([[f[0], f[3], f[4]], [f[3], f[1], f[5]], [f[4], f[5], f[2]]]))
            else:
                body_inertia[b], body_iquat[b] = given["diag"], given["quat"]
            continue
        if c.inertiafromgeom == "false":
            continue                    # (no <inertial>: massless -- refused below if the body can move)
        parts = []
        for gi in c.bodies[b]["geoms"]:
            mass, inertia = _geom_mass_inertia(c.geoms[gi])
            if mass > 0:
                parts.append((mass, inertia, c.geoms[gi]["pos"], c.geoms[gi]["quat"]))
        if not parts:
            continue
        if len(parts) == 1:
            body_mass[b], body_inertia[b], body_ipos[b], body_iquat[b] = parts[0]
            continue
        total = sum(p[0] for p in parts)
        com = sum(p[0] * p[2] for p in parts) / total
        tensor = np.zeros((3, 3))
        for mass, inertia, pos, quat in parts:
            rot = quat_to_mat(quat)
            d = pos - com
            tensor += rot @ np.diag(inertia) @ rot.T + mass * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
        evals, iquat = principal(tensor)
        body_mass[b], body_ipos[b], body_inertia[b], body_iquat[b] = total, com, evals, iquat
    for b in range(1, nbody):
        # (MuJoCo's compiler refuses the same: "mass and inertia of moving bodies must be larger than mjMINVAL")
        if body_jntnum[b] > 0 and (body_mass[b] < MINVAL or np.min(body_inertia[b]) < MINVAL) and body_subtree_massless(b, body_parentid, body_mass):
            raise ValueError(f"body '{c.bodies[b]['name']}' moves but it and everything it carries have no mass")
    body_subtreemass = body_mass.copy()
    for b in range(nbody - 1, 0, -1):
        body_subtreemass[body_parentid[b]] += body_subtreemass[b]
    A.update(body_mass=body_mass, body_ipos=body_ipos, body_iquat=body_iquat, body_inertia=body_inertia,
             body_subtreemass=body_subtreemass)

    # ---- sites, cameras
    ns, nc = m.nsite, m.ncam
    A["site_bodyid"] = np.array([s["body"] for s in c.sites], np.int32)
    A["site_pos"] = np.array([s["pos"] for s in c.sites], np.float64).reshape(ns, 3)
    A["site_quat"] = np.array([s["quat"] for s in c.sites], np.float64).reshape(ns, 4)
    A["site_size"] = np.array([s["size"] for s in c.sites], np.float64).reshape(ns, 3)
    A["cam_bodyid"] = np.array([s["body"] for s in c.cams], np.int32)
    A["cam_pos"] = np.array([s["pos"] for s in c.cams], np.float64).reshape(nc, 3)
    A["cam_quat"] = np.array([s["quat"] for s in c.cams], np.float64).reshape(nc, 4)
    A["cam_fovy"] = np.array([s["fovy"] for s in c.cams], np.float64)
    # ---- lights (rendering only): the scene's <light> elements and the headlight of <visual>
    nl = m.nlight = len(c.lights)
    A["light_bodyid"] = np.array([s["body"] for s in c.lights], np.int32)
    A["light_directional"] = np.array([s["directional"] for s in c.lights], np.int32)
    A["light_castshadow"] = np.array([s["castshadow"] for s in c.lights], np.int32)
    for key, width in (("pos", 3), ("dir", 3), ("attenuation", 3), ("ambient", 3), ("diffuse", 3), ("specular", 3)):
        A["light_" + key] = np.array([s[key] for s in c.lights], np.float64).reshape(nl, width)
    A["light_cutoff"] = np.array([s["cutoff"] for s in c.lights], np.float64)
    A["light_exponent"] = np.array([s["exponent"] for s in c.lights], np.float64)
    h = c.headlight
    A["headlight"] = np.concatenate([[h["active"]], h["ambient"], h["diffuse"], h["specular"]]).astype(np.float64)

    m.names = dict(body=[b["name"] for b in c.bodies], joint=[j["name"] for j in c.joints],
                   geom=[g["name"] for g in c.geoms], site=[s["name"] for s in c.sites],
                   camera=[s["name"] for s in c.cams], actuator=[], sensor=[])
    # a camera that is not "fixed" in its body (Ant.xml's mode="trackcom") follows a rule the ray caster does not
    # implement: the level compiles (physics never reads a camera), drawing that camera is refused (get_camera_data)
    m.camera_mode = [s["mode"] for s in c.cams]

    # ---- actuators (motors)
    act = c.root.find("actuator")
    motors = []
    if act is not None:
        for el in act:
            a = c.defaults.apply(el, None)
            if el.tag not in ("motor", "general"):
                raise ValueError(f"actuator <{el.tag}> is outside the supported MJCF subset")
            jid = m.names["joint"].index(a["joint"])
            gear = _vec(a.get("gear"), 6, [1, 0, 0, 0, 0, 0])
            motors.append(dict(name=a.get("name", ""), dof=int(jnt_dofadr[jid]), jnt=jid, gear=gear[0],
                               limited=a.get("ctrllimited", "false") == "true",
                               range=_vec(a.get("ctrlrange"), 2, [0, 0])))
    m.nu = len(motors)
    A["act_dofid"] = np.array([a["dof"] for a in motors], np.int32)
    A["act_jntid"] = np.array([a["jnt"] for a in motors], np.int32)
    A["act_gear"] = np.array([a["gear"] for a in motors], np.float64)
    A["act_ctrllimited"] = np.array([int(a["limited"]) for a in motors], np.int32)
    A["act_ctrlrange"] = np.array([a["range"] for a in motors], np.float64).reshape(m.nu, 2)
    m.names["actuator"] = [a["name"] for a in motors]
    # the single actuator driving each dof (-1 none, -2 several: the kernel then scans the actuator list)
    dof_actid = np.full(nv, -1, np.int32)
    for u, a in enumerate(motors):
        dof_actid[a["dof"]] = u if dof_actid[a["dof"]] == -1 else -2
    A["dof_actid"] = dof_actid

    # ---- sensors (the subset the levels use; all are attached to a site)
    sens = c.root.find("sensor")
    s_type, s_objid, s_dim, s_adr, s_cutoff = [], [], [], [], []
    adr = 0
    if sens is not None:
        for el in sens:
            if el.tag not in SENSOR_TYPES:
                raise ValueError(f"sensor <{el.tag}> is outside the supported MJCF subset")
            stype, dim = SENSOR_TYPES[el.tag]
            site_name = el.attrib.get("site", el.attrib.get("objname"))
            if el.tag.startswith("frame") and el.attrib.get("objtype", "site") != "site":
                raise ValueError("frame sensors are supported on sites only")
            # (a touch sensor's active zone is its site's SHAPE -- the kernels test a sphere of radius size[0]; a frame or a
            # rangefinder reads the site's frame only, whatever its shape)
            if el.tag == "touch" and c.sites[m.names["site"].index(site_name)]["type"] != "sphere":
                _refuse(f'a touch sensor on a site of type "{c.sites[m.names["site"].index(site_name)]["type"]}" (sphere sites only)')
            s_type.append(stype)
            s_objid.append(m.names["site"].index(site_name))
            s_dim.append(dim)
            s_adr.append(adr)
            s_cutoff.append(float(el.attrib.get("cutoff", 0.0)))
            m.names["sensor"].append(el.attrib.get("name", ""))
            adr += dim
    m.nsensor, m.nsensordata = len(s_type), adr
    A["sensor_type"] = np.array(s_type, np.int32)
    A["sensor_objid"] = np.array(s_objid, np.int32)
    A["sensor_dim"] = np.array(s_dim, np.int32)
    A["sensor_adr"] = np.array(s_adr, np.int32)
    A["sensor_cutoff"] = np.array(s_cutoff, np.float64)

    # ---- candidate collision pairs (static filter), ordered by (geom1, geom2) before type swap
    pairs = []
    gt, gb = A["geom_type"], A["geom_bodyid"]
    # <contact><exclude body1 body2/>: no geom of the one body collides with a geom of the other
    excluded = set()
    for con in c.root.findall("contact"):
        for el in con:
            ids = []
            for key in ("body1", "body2"):
                name = el.attrib.get(key)
                if name in folded:
                    ids.append(0)
                elif name in m.names["body"]:
                    ids.append(m.names["body"].index(name))
                else:
                    raise KeyError(f"<exclude>: no body named '{name}'")
            excluded.add((min(ids), max(ids)))
    for g1 in range(ngeom):
        for g2 in range(g1 + 1, ngeom):
            b1, b2 = gb[g1], gb[g2]
            if b1 == b2 or (min(int(b1), int(b2)), max(int(b1), int(b2))) in excluded:
                continue
            g_a, g_b = c.geoms[g1], c.geoms[g2]
            if not ((g_a["contype"] & g_b["conaffinity"]) or (g_b["contype"] & g_a["conaffinity"])):
                continue
            w1, w2 = body_weldid[b1], body_weldid[b2]
            if w1 == w2:
                continue
            wp1, wp2 = body_weldid[body_parentid[w1]], body_weldid[body_parentid[w2]]
            if w1 != 0 and w2 != 0 and (w1 == wp2 or w2 == wp1):
                continue
            a, b = (g1, g2) if gt[g1] <= gt[g2] else (g2, g1)
            if (int(gt[a]), int(gt[b])) not in SUPPORTED_PAIRS:
                raise ValueError(f"collision pair types {gt[a]},{gt[b]} are outside the supported subset")
            pairs.append((a, b))
    rb, mg = A["geom_rbound"], A["geom_margin"]
    pairs, chunk_info, tp_root, tp_reach = _pair_layout(m, A, pairs, broad_cull)
    m.npair, m.nchunk, m.ntp = len(pairs), len(chunk_info), len(tp_reach)
    kinds = [ci & 255 for ci in chunk_info]
    m.nchunk_plane, m.nchunk_box, m.nchunk_boxbox = kinds.count(PAIR_PLANE), kinds.count(PAIR_BOX), kinds.count(PAIR_BOXBOX)
    assert kinds == sorted(kinds)               # the segments come in the order of the kinds' codes
    A["chunk_info"] = np.array(chunk_info, np.int32)
    A["tp_root"] = np.array(tp_root, np.int32).reshape(-1)
    A["tp_reach"] = np.array(tp_reach, np.float64)
    real = [pr for pr in pairs if pr is not None]
    A["pair_geom"] = np.array([pr if pr is not None else (-1, -1) for pr in pairs], np.int32).reshape(m.npair, 2)
    # per-pair constants (0 in the padding entries); `val(f)` evaluates f(a, b) on the real pairs
    def val(f):
        return np.array([f(*pr) if pr is not None else 0.0 for pr in pairs], np.float64)
    # broad-phase constants per pair: contact margin and the bounding-sphere reach (plane pairs: the other geom's)
    A["pair_margin"] = val(lambda a, b: max(mg[a], mg[b]))
    # what a contact of the pair takes from its two geoms, so that the narrow phase reads one record per pair instead of
    # following the geom ids: gap and sliding friction, both the larger of the two (MuJoCo's mixing rule for geom pairs)
    gp, fr = A["geom_gap"], np.asarray(A["geom_friction"], np.float64).reshape(-1, 3)
    A["pair_gap"] = val(lambda a, b: max(gp[a], gp[b]))
    A["pair_mu"] = val(lambda a, b: max(fr[a, 0], fr[b, 0]))
    # reach of the broad-phase test: plane pairs and (sphere|capsule)-box pairs test ONE bounding sphere against the
    # plane / the box itself, every other pair tests the two bounding spheres against each other
    def reach(a, b):
        if gt[a] == GEOM_PLANE:
            return rb[b]
        if gt[b] == GEOM_BOX and gt[a] != GEOM_BOX:
            return rb[a]
        return rb[a] + rb[b]
    A["pair_bound"] = val(lambda a, b: reach(a, b) + max(mg[a], mg[b]))

    # caps per env copy (MuJoCo's <size nconmax njmax>): by default room for 6 contacts per kinematic tree and
    # one limit row per limited joint plus a 4-row pyramid per contact.  The caps size the constraint block of the
    # copy's LDS image (16 doubles of Jacobian per row), i.e. how many copies a CU holds at once; a step that runs
    # into one is counted (mjrl_cap_overflows) and the config keys ``nconmax`` / ``njmax`` raise them.
    m.nconmax = int(nconmax) if nconmax is not None else min(64, 6 * max(m.ntree, 1))
    n_limited = int(np.sum(A["jnt_limited"]))
    m.njmax = int(njmax) if njmax is not None else n_limited + 4 * m.nconmax

    _set_const(m)
    # <statistic meaninertia="..."/> overrides the computed value (it scales the solver's stop test); the section's other
    # attributes -- extent, center, meansize, meanmass -- scale the visualisation only
    stat = c.root.find("statistic")
    if stat is not None and "meaninertia" in stat.attrib:
        m.meaninertia = float(stat.attrib["meaninertia"])
    _kernel_schedules(m, lane_map)
    return m


PAIR_PLANE, PAIR_BOX, PAIR_BOXBOX, PAIR_SPHERES = range(4)      # the broad-phase test of a chunk of candidate pairs


def _pair_layout(m: Model, A: dict, pairs: list, broad_cull: bool):
    """Order of the candidate pairs (= the order of the contacts, shared with the oracle through the blob).

    The broad phase tests 64 pairs at a time, and what a chunk costs is set by the KINDS of test in it (every kind
    present runs for the whole wave) and by whether it has to run at all.  So the list is laid out in segments of one
    kind each, padded to whole chunks with empty entries (``None``): plane pairs, (sphere | capsule)-box pairs,
    box-box pairs, bounding-sphere pairs inside one tree or against static geoms, then one block of bounding-sphere
    pairs per pair of kinematic trees.  A block is skipped when the two trees' bounding spheres (root body position,
    a reach that no joint configuration can exceed) are further apart than any contact margin of the block reaches:
    the pairs of a skipped block would all fail their own tests, so nothing changes but the work.  Inside a segment the
    pairs keep their (geom1, geom2) order.

    Returns (pairs with padding, chunk_info per chunk = kind | (tree pair + 1) << 8, tp_root, tp_reach)."""
    gt, gb = A["geom_type"], A["geom_bodyid"]
    tree_of = [int(A["body_treeid"][gb[g]]) for g in range(len(gt))]
    ntree = int(m.ntree)
    # a bound on the distance from a tree's root body origin to any point of its geoms: offsets along the body chain
    # keep their length under rotation, a slide joint adds at most the larger end of its range
    reach_of = np.full(max(ntree, 1), np.inf)
    if broad_cull and 2 <= ntree <= 11:
        roots = [int(r) for r in A["tree_rootbody"]]
        slide = np.zeros(m.nbody)
        ok = True
        for j in range(m.njnt):
            if A["jnt_type"][j] == JNT_SLIDE:
                if not A["jnt_limited"][j]:
                    ok = False
                slide[int(A["jnt_bodyid"][j])] += float(np.max(np.abs(A["jnt_range"][j]))) + float(np.linalg.norm(A["jnt_pos"][j]))
            elif A["jnt_type"][j] in (JNT_HINGE, JNT_BALL):
                # rotation about an anchor off the body origin moves the origin on a sphere of that radius
                slide[int(A["jnt_bodyid"][j])] += 2.0 * float(np.linalg.norm(A["jnt_pos"][j]))
        if ok:
            reach_of[:] = 0.0
            for g in range(len(gt)):
                t = tree_of[g]
                if t < 0:
                    continue
                d = float(np.linalg.norm(A["geom_pos"][g])) + float(A["geom_rbound"][g])
                b = int(gb[g])
                while b != roots[t]:
                    d += float(np.linalg.norm(A["body_pos"][b])) + slide[b]
                    b = int(A["body_parentid"][b])
                reach_of[t] = max(reach_of[t], d)
    blocks = np.all(np.isfinite(reach_of)) and broad_cull and 2 <= ntree <= 11
    segs = {("P",): [], ("B",): [], ("BB",): [], ("S",): []}
    order = [("P",), ("B",), ("BB",), ("S",)]
    for a, b in pairs:
        ta, tb = int(gt[a]), int(gt[b])
        if ta == GEOM_PLANE:
            key = ("P",)
        elif tb == GEOM_BOX and ta != GEOM_BOX:
            key = ("B",)
        elif ta == GEOM_BOX and tb == GEOM_BOX:
            key = ("BB",)
        else:
            s1, s2 = sorted((tree_of[a], tree_of[b]))
            key = ("S", s1, s2) if (blocks and s1 >= 0 and s1 != s2) else ("S",)
            if key not in segs:
                segs[key] = []
                order.append(key)
        segs[key].append((int(a), int(b)))
    order = order[:4] + sorted(order[4:])
    kind_of = {"P": PAIR_PLANE, "B": PAIR_BOX, "BB": PAIR_BOXBOX, "S": PAIR_SPHERES}
    mg = A["geom_margin"]
    out, chunk_info, tp_root, tp_reach = [], [], [], []
    for key in order:
        seg = segs[key]
        if not seg:
            continue
        tp = 0
        if len(key) == 3:
            tp_root += [int(A["tree_rootbody"][key[1]]), int(A["tree_rootbody"][key[2]])]
            margin = max(max(mg[a], mg[b]) for a, b in seg)
            tp_reach.append(float(np.nextafter(reach_of[key[1]] + reach_of[key[2]] + margin, np.inf)) * (1.0 + 1e-12))
            tp = len(tp_reach)
        nch = (len(seg) + 63) // 64
        out += seg + [None] * (64 * nch - len(seg))
        chunk_info += [kind_of[key[0]] | (tp << 8)] * nch
    return out, chunk_info, tp_root, tp_reach


def _kernel_schedules(m: Model, lane_map: bool = True):
    """Lane schedules the step kernel replays instead of decoding the tree structure on the fly.

    * ``factor_sched[pass][kk][lane]``: the L'DL elimination eliminates the kk-th dof of up to two trees per pass
      (32 lanes each); a lane owns one (ancestor a, offset t) pair of the pivot row.  Packed word:
      bits 0-9 address of the pivot diagonal M[k][k] (L[k][i] sits at that + 1 + a, M[k][j] at that + 1 + a + t),
      bits 10-19 address of M[i][j] to update, 20-22 a, 23-25 t, bit 26 pair valid.
    * tree-row lane map (``rowmap`` = 1 when every tree has <= 16 dofs and there are <= 4 trees): tree t owns lanes
      16t..16t+15, ``row_dof[lane]`` is the lane's dof (or -1).  ``solve_b[kk][lane]`` / ``solve_f[kk][lane]`` hold
      the factor entries a lane multiplies in step kk of the backward / forward triangular solve (or -1).
    """
    A = m.arrays
    nv, ntree = m.nv, m.ntree
    Madr, depth, colid = A["dof_Madr"], A["dof_depth"], A["M_colid"]
    tadr, tnum = A["tree_dofadr"], A["tree_dofnum"]
    groups = 2 if ntree > 1 else 1
    width = 64 // groups
    npass = (ntree + groups - 1) // groups if ntree else 0
    sched = np.zeros((max(npass, 1), max(m.maxtreedof, 1), 64), np.int64)
    for ps in range(npass):
        for g in range(groups):
            tree = ps * groups + g
            if tree >= ntree:
                continue
            for kk in range(int(tnum[tree])):
                k = int(tadr[tree]) + kk
                D, kkadr = int(depth[k]), int(Madr[k])
                pairs = [(a, t) for a in range(D) for t in range(D - a)]
                if len(pairs) > width:
                    raise ValueError("a dof chain is too deep for the factorisation schedule (more than %d pairs)" % width)
                for lane in range(width):
                    word = kkadr
                    if lane < len(pairs):
                        a, t = pairs[lane]
                        ki = kkadr + 1 + a
                        ij = int(Madr[colid[ki]])
                        word |= ((ij + t) << 10) | (a << 20) | (t << 23) | (1 << 26)
                        if ki + t >= 1024 or ij + t >= 1024 or a > 7 or t > 7:
                            raise ValueError("sparse inertia matrix too large for the packed schedule (nM >= 1024)")
                    sched[ps, kk, g * width + lane] = word
    m.npass = npass
    A["factor_sched"] = sched.astype(np.uint32).view(np.int32).reshape(-1)
    m.nfactor = int(A["factor_sched"].size)

    rowmap = int(lane_map and ntree >= 1 and ntree <= 4 and m.maxtreedof <= 16)
    m.rowmap = rowmap
    row_dof = np.full(64, -1, np.int32)
    solve_b = np.full((16, 64), -1, np.int32)
    solve_f = np.full((16, 64), -1, np.int32)
    if rowmap:
        for t in range(ntree):
            for loc in range(int(tnum[t])):
                row_dof[16 * t + loc] = int(tadr[t]) + loc
        for lane in range(64):
            d = int(row_dof[lane])
            if d < 0:
                continue
            t = lane // 16
            anc = [int(colid[Madr[d] + s]) for s in range(1, int(depth[d]) + 1)]      # proper ancestors of d, walking up
            for kk in range(int(tnum[t])):
                k = int(tadr[t]) + kk
                # backward step kk: x[d] -= L[k][d] * x[k] when d is a proper ancestor of k
                kanc = [int(colid[Madr[k] + s]) for s in range(1, int(depth[k]) + 1)]
                if d in kanc:
                    solve_b[kk, lane] = int(Madr[k]) + 1 + kanc.index(d)
                # forward step kk: x[d] -= L[d][k] * x[k] when k is a proper ancestor of d
                if k in anc:
                    solve_f[kk, lane] = int(Madr[d]) + 1 + anc.index(k)
    A["row_dof"], A["solve_b"], A["solve_f"] = row_dof, solve_b.reshape(-1), solve_f.reshape(-1)
    # tree of every body's dofs as seen by a constraint row (static bodies: -1) is body_treeid; dof -> lane
    dof_lane = np.full(max(nv, 1), -1, np.int32)
    for lane in range(64):
        if row_dof[lane] >= 0:
            dof_lane[row_dof[lane]] = lane
    A["dof_lane"] = dof_lane[:nv] if nv else np.zeros(0, np.int32)
    # Structure tables the row build walks with data-dependent indices; the kernel keeps them in LDS as 16-bit words.
    # Order (see mjrl_step.h, TabOff): M_colid[nM], dof_Madr[nv], dof_depth[nv], dof_treeid[nv], body_lastdof+1[nbody],
    # body_treeid+1[nbody], geom_bodyid[ngeom], geom_type[ngeom], geom_condim[ngeom]
    A["lds_tab"] = np.concatenate([
        colid, Madr, depth, A["dof_treeid"], A["body_lastdof"] + 1, A["body_treeid"] + 1,
        A["geom_bodyid"], A["geom_type"], A["geom_condim"]]).astype(np.int32)
    m.ntab = int(A["lds_tab"].size)
    if A["lds_tab"].size and (A["lds_tab"].max() > 65535 or A["lds_tab"].min() < 0):
        raise ValueError("structure table entry does not fit 16 bits")
    # broad-phase word per candidate pair: geom1 | geom2 << 8 | type1 << 16 | type2 << 20, and the reach as float32
    # (rounded up: the broad phase only has to be conservative)
    pg = A["pair_geom"]
    gt = A["geom_type"]
    # bit 24: the geom whose FRAME the broad-phase test needs (the plane of a plane pair, the box of a sphere|capsule-box
    # pair) never rotates and has the identity orientation -- a static body chain and a geom with exactly identity
    # quaternions (the arena's floor and walls).  Its rotation matrix is then exactly the identity and the test skips
    # building and applying it; the results are the same bits.
    ident = np.array([1.0, 0.0, 0.0, 0.0])
    def fixed_identity(g):
        b = int(A["geom_bodyid"][g])
        if not np.array_equal(A["geom_quat"][g], ident):
            return False
        while b > 0:
            if m.body_jntnum[b] > 0 or not np.array_equal(m.body_quat[b], ident):
                return False
            b = int(m.body_parentid[b])
        return True
    def frame_flag(a, b):
        if gt[a] == GEOM_PLANE:
            return fixed_identity(a)
        if gt[b] == GEOM_BOX and gt[a] != GEOM_BOX:
            return fixed_identity(b)
        return False
    # bits 25-27: log2 of the narrow-phase work items of the pair (1, 2, 4, 8 or 16), bit 28: a capsule-capsule pair (its
    # four items are needed only when the capsules are parallel; otherwise one).  Bit 31: a padding entry, never a pair.
    items_log2 = {(GEOM_PLANE, GEOM_CAPSULE): 1, (GEOM_PLANE, GEOM_BOX): 3, (GEOM_CAPSULE, GEOM_CAPSULE): 2,
                  (GEOM_CAPSULE, GEOM_BOX): 1, (GEOM_BOX, GEOM_BOX): 4}
    def word_of(a, b):
        if a < 0:
            return -(1 << 31)
        ta, tb = int(gt[a]), int(gt[b])
        w = int(a) | (int(b) << 8) | (ta << 16) | (tb << 20) | (int(frame_flag(a, b)) << 24)
        w |= items_log2.get((ta, tb), 0) << 25
        w |= int(ta == GEOM_CAPSULE and tb == GEOM_CAPSULE) << 28
        return w
    A["pair_word"] = np.array([word_of(int(a), int(b)) for a, b in pg], np.int64).astype(np.int32)
    A["pair_reach"] = (np.nextafter(A["pair_bound"].astype(np.float32), np.float32(np.inf))).view(np.int32) \
        if len(pg) else np.zeros(0, np.int32)


# ----------------------------------------------------------------------------- constants at qpos0
def kinematics_numpy(m: Model, qpos):
    """Body frames (xpos, xquat) for ``qpos`` -- plain numpy, used for the qpos0 constants only."""
    nb = m.nbody
    xpos, xquat = np.zeros((nb, 3)), np.tile([1.0, 0, 0, 0], (nb, 1))
    for b in range(1, nb):
        p = m.body_parentid[b]
        rot_p = quat_to_mat(xquat[p])
        pos = xpos[p] + rot_p @ m.body_pos[b]
        quat = quat_mul(xquat[p], m.body_quat[b])
        for k in range(m.body_jntnum[b]):
            j = m.body_jntadr[b] + k
            qa = m.jnt_qposadr[j]
            if m.jnt_type[j] == JNT_FREE:
                pos = np.array(qpos[qa:qa + 3])
                quat = np.array(qpos[qa + 3:qa + 7])
                quat = quat / np.linalg.norm(quat)
            else:
                anchor = pos + quat_to_mat(quat) @ m.jnt_pos[j]
                if m.jnt_type[j] == JNT_HINGE:
                    quat = quat_mul(quat, axis_angle_quat(m.jnt_axis[j], qpos[qa] - m.qpos0[qa]))
                    pos = anchor - quat_to_mat(quat) @ m.jnt_pos[j]
                else:
                    pos = pos + quat_to_mat(quat) @ m.jnt_axis[j] * (qpos[qa] - m.qpos0[qa])
        xpos[b], xquat[b] = pos, quat / np.linalg.norm(quat)
    return xpos, xquat


def body_jacobian_numpy(m: Model, xpos, xquat, body, point):
    """6 x nv Jacobian (translational rows first) of ``point`` moving with ``body``."""
    jac = np.zeros((6, m.nv))
    d = m.body_lastdof[body]
    while d >= 0:
        b, j = m.dof_bodyid[d], m.dof_jntid[d]
        rot = quat_to_mat(xquat[b])
        if m.jnt_type[j] == JNT_FREE:
            k = d - m.jnt_dofadr[j]
            if k < 3:
                jac[k, d] = 1.0
            else:
                axis = rot[:, k - 3]
                jac[3:, d] = axis
                jac[:3, d] = np.cross(axis, point - xpos[b])
        else:
            axis = rot @ m.jnt_axis[j]
            if m.jnt_type[j] == JNT_HINGE:
                anchor = xpos[b] + rot @ m.jnt_pos[j]
                jac[3:, d] = axis
                jac[:3, d] = np.cross(axis, point - anchor)
            else:
                jac[:3, d] = axis
        d = m.dof_parentid[d]
    return jac


def mass_matrix_numpy(m: Model, qpos):
    xpos, xquat = kinematics_numpy(m, qpos)
    M = np.diag(m.dof_armature.astype(np.float64))
    for b in range(1, m.nbody):
        if m.body_mass[b] <= 0 or m.body_lastdof[b] < 0:
            continue
        rot = quat_to_mat(xquat[b])
        com = xpos[b] + rot @ m.body_ipos[b]
        jac = body_jacobian_numpy(m, xpos, xquat, b, com)
        irot = rot @ quat_to_mat(m.body_iquat[b])
        inertia_w = irot @ np.diag(m.body_inertia[b]) @ irot.T
        M += m.body_mass[b] * jac[:3].T @ jac[:3] + jac[3:].T @ inertia_w @ jac[3:]
    return M, xpos, xquat


def _set_const(m: Model):
    """``body_invweight0``, ``dof_invweight0``, ``meaninertia`` at ``qpos0``."""
    A = m.arrays
    inv_b = np.zeros((m.nbody, 2))
    inv_d = np.zeros(m.nv)
    if m.nv:
        M, xpos, xquat = mass_matrix_numpy(m, m.qpos0)
        Minv = np.linalg.inv(M)
        m.meaninertia = float(np.mean(np.diag(M)))
        for b in range(1, m.nbody):
            if m.body_lastdof[b] < 0:
                continue
            com = xpos[b] + quat_to_mat(xquat[b]) @ m.body_ipos[b]
            jac = body_jacobian_numpy(m, xpos, xquat, b, com)
            a = jac @ Minv @ jac.T
            inv_b[b, 0] = (a[0, 0] + a[1, 1] + a[2, 2]) / 3.0
            inv_b[b, 1] = (a[3, 3] + a[4, 4] + a[5, 5]) / 3.0
        for j in range(m.njnt):
            d = m.jnt_dofadr[j]
            if m.jnt_type[j] == JNT_FREE:
                inv_d[d:d + 3] = np.mean(np.diag(Minv)[d:d + 3])
                inv_d[d + 3:d + 6] = np.mean(np.diag(Minv)[d + 3:d + 6])
            else:
                inv_d[d] = Minv[d, d]
    A["body_invweight0"] = inv_b
    A["dof_invweight0"] = inv_d
