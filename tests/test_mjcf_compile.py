"""Known answers for the model compiler (``mjcf.py``) -- the one piece that sits under BOTH sides of every parity test
(oracle and kernel read the same packed blob), so nothing else would notice an error in it.

The reference hands its XML to ``mujoco.MjModel.from_xml_path`` (mujoco_parent.py:126); what that call is documented to
compute -- inertia inferred from geoms of uniform density, the composite body's principal frame, degrees, default
classes, the constants at ``qpos0`` -- is checked here against closed forms and against quadrature written without the
compiler's formulas.  The second half checks that a level outside the implemented subset is REFUSED by name rather than
simulated without the feature."""
import math

import numpy as np
import pytest
from scipy import integrate

from mjrl_amd import mjcf


def compile_xml(body: str, head: str = "", tail: str = ""):
    return mjcf.compile_mjcf_string(f"<mujoco>{head}<worldbody>{body}</worldbody>{tail}</mujoco>")


# ---------------------------------------------------------------------------------------- solids of revolution by quadrature
def solid_of_revolution(radius_of_z, z0, z1, rho):
    """Mass, axial and lateral moment (about the origin) of the solid swept by discs of radius R(z)."""
    mass = integrate.quad(lambda z: rho * math.pi * radius_of_z(z) ** 2, z0, z1, epsabs=1e-14, epsrel=1e-13)[0]
    axial = integrate.quad(lambda z: rho * math.pi * radius_of_z(z) ** 4 / 2, z0, z1, epsabs=1e-14, epsrel=1e-13)[0]
    lateral = integrate.quad(lambda z: rho * math.pi * (radius_of_z(z) ** 4 / 4 + radius_of_z(z) ** 2 * z * z), z0, z1,
                             epsabs=1e-14, epsrel=1e-13)[0]
    return mass, axial, lateral


@pytest.mark.parametrize("density", [5.0, None])                    # the levels' density and MuJoCo's default (1000)
def test_sphere_mass_and_inertia(density):
    rho = 1000.0 if density is None else density
    attr = "" if density is None else f'density="{density}"'
    r = 0.25
    m = compile_xml(f'<body pos="0 0 1"><freejoint/><geom type="sphere" size="{r}" {attr}/></body>')
    mass = 4.0 / 3.0 * math.pi * r ** 3 * rho
    assert m.body_mass[1] == pytest.approx(mass, rel=1e-14)
    assert np.allclose(m.body_inertia[1], 2.0 / 5.0 * mass * r * r, rtol=1e-14)
    qm, qa, ql = solid_of_revolution(lambda z: math.sqrt(max(r * r - z * z, 0.0)), -r, r, rho)
    assert m.body_mass[1] == pytest.approx(qm, rel=1e-10)
    assert np.allclose(m.body_inertia[1], [ql, ql, qa], rtol=1e-10)


@pytest.mark.parametrize("density", [5.0, None])
def test_capsule_mass_and_inertia_against_quadrature(density):
    rho = 1000.0 if density is None else density
    attr = "" if density is None else f'density="{density}"'
    r, half = 0.08, 0.2
    m = compile_xml(f'<body pos="0 0 1"><freejoint/><geom type="capsule" size="{r} {half}" {attr}/></body>')

    def radius(z):
        over = abs(z) - half
        return r if over <= 0 else math.sqrt(max(r * r - over * over, 0.0))
    qm, qa, ql = solid_of_revolution(radius, -half - r, half + r, rho)
    assert m.body_mass[1] == pytest.approx(rho * (math.pi * r * r * 2 * half + 4.0 / 3.0 * math.pi * r ** 3), rel=1e-14)
    assert m.body_mass[1] == pytest.approx(qm, rel=1e-10)
    assert np.allclose(m.body_inertia[1], [ql, ql, qa], rtol=1e-9)
    # the hemispherical caps in closed form: own moment 83/320 m r^2 about their centroid, 3r/8 beyond the flat face
    m_cyl, m_cap = rho * math.pi * r * r * 2 * half, rho * 2.0 / 3.0 * math.pi * r ** 3
    lateral = m_cyl * (3 * r * r + 4 * half * half) / 12 + 2 * m_cap * (83.0 / 320 * r * r + (half + 3 * r / 8) ** 2)
    assert m.body_inertia[1][0] == pytest.approx(lateral, rel=1e-13)
    assert m.body_inertia[1][2] == pytest.approx(m_cyl * r * r / 2 + 2 * m_cap * 0.4 * r * r, rel=1e-13)


@pytest.mark.parametrize("density", [5.0, None])
def test_box_mass_and_inertia(density):
    rho = 1000.0 if density is None else density
    attr = "" if density is None else f'density="{density}"'
    a, b, c = 0.3, 0.2, 0.1                                              # half sizes
    m = compile_xml(f'<body pos="0 0 1"><freejoint/><geom type="box" size="{a} {b} {c}" {attr}/></body>')
    mass = rho * 8 * a * b * c
    assert m.body_mass[1] == pytest.approx(mass, rel=1e-14)
    # a cuboid with full edges 2a, 2b, 2c: I_x = m ((2b)^2 + (2c)^2) / 12
    expect = mass / 12 * np.array([4 * b * b + 4 * c * c, 4 * a * a + 4 * c * c, 4 * a * a + 4 * b * b])
    assert np.allclose(m.body_inertia[1], expect, rtol=1e-14)


def test_explicit_geom_mass_overrides_the_density():
    m = compile_xml('<body pos="0 0 1"><freejoint/><geom type="sphere" size="0.1" mass="3.5" density="17"/></body>')
    assert m.body_mass[1] == pytest.approx(3.5, rel=1e-14)
    assert np.allclose(m.body_inertia[1], 0.4 * 3.5 * 0.01, rtol=1e-14)


def test_fromto_capsule_frame_and_size():
    m = compile_xml('<body pos="0 0 1"><freejoint/>'
                    '<geom type="capsule" fromto="0.1 0 0  0.1 0.6 0" size="0.05"/></body>')
    assert np.allclose(m.geom_pos[0], [0.1, 0.3, 0.0])
    assert np.allclose(m.geom_size[0][:2], [0.05, 0.3])
    axis = mjcf.quat_to_mat(m.geom_quat[0])[:, 2]
    assert np.allclose(np.abs(axis), [0, 1, 0], atol=1e-15)             # the capsule's z axis lies along from -> to


def inertia_tensor(m, b):
    rot = mjcf.quat_to_mat(m.body_iquat[b])
    return rot @ np.diag(m.body_inertia[b]) @ rot.T


def test_two_geom_body_against_the_parallel_axis_theorem():
    # a sphere and a box, neither at the body origin, the box turned 30 degrees about z
    r, rho = 0.1, 5.0
    a, b, c = 0.2, 0.05, 0.1
    ps, pb = np.array([0.3, 0.0, 0.1]), np.array([-0.1, 0.2, 0.0])
    m = compile_xml(f'<body pos="0 0 1"><freejoint/>'
                    f'<geom type="sphere" size="{r}" pos="{ps[0]} {ps[1]} {ps[2]}"/>'
                    f'<geom type="box" size="{a} {b} {c}" pos="{pb[0]} {pb[1]} {pb[2]}" euler="0 0 30"/></body>',
                    head=f'<default><geom density="{rho}"/></default>')
    ms, mb = rho * 4 / 3 * math.pi * r ** 3, rho * 8 * a * b * c
    total = ms + mb
    com = (ms * ps + mb * pb) / total
    assert m.body_mass[1] == pytest.approx(total, rel=1e-14)
    assert np.allclose(m.body_ipos[1], com, rtol=1e-13)
    cz, sz = math.cos(math.radians(30)), math.sin(math.radians(30))
    rot = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    box = rot @ np.diag(mb / 3 * np.array([b * b + c * c, a * a + c * c, a * a + b * b])) @ rot.T

    def shifted(tensor, mass, d):
        return tensor + mass * (d @ d * np.eye(3) - np.outer(d, d))
    expect = shifted(np.eye(3) * 0.4 * ms * r * r, ms, ps - com) + shifted(box, mb, pb - com)
    assert np.allclose(inertia_tensor(m, 1), expect, rtol=1e-12, atol=1e-16)
    # principal moments in descending order, a right-handed frame
    assert np.all(np.diff(m.body_inertia[1]) <= 0)
    assert np.allclose(np.sort(m.body_inertia[1]), np.linalg.eigvalsh(expect), rtol=1e-12)
    assert np.linalg.det(mjcf.quat_to_mat(m.body_iquat[1])) == pytest.approx(1.0, abs=1e-13)


def test_dumbbell_of_two_equal_spheres():
    r, d, rho = 0.05, 0.4, 1000.0
    m = compile_xml(f'<body pos="0 0 1"><freejoint/><geom type="sphere" size="{r}" pos="{d} 0 0"/>'
                    f'<geom type="sphere" size="{r}" pos="{-d} 0 0"/></body>')
    ms = rho * 4 / 3 * math.pi * r ** 3
    across = 2 * (0.4 * ms * r * r + ms * d * d)
    assert np.allclose(m.body_ipos[1], 0, atol=1e-16)
    assert np.allclose(m.body_inertia[1], [across, across, 2 * 0.4 * ms * r * r], rtol=1e-13)
    assert np.allclose(inertia_tensor(m, 1), np.diag([2 * 0.4 * ms * r * r, across, across]), rtol=1e-12, atol=1e-15)


# ---------------------------------------------------------------------------------------- angles, defaults, qpos0
def test_hinge_range_is_degrees_unless_the_compiler_says_radian():
    body = ('<body pos="0 0 1"><joint type="hinge" axis="0 1 0" limited="true" range="30 70"/>'
            '<geom type="sphere" size="0.1"/></body>')
    m = compile_xml(body)
    assert np.allclose(m.jnt_range[0], [math.pi / 6, 70 * math.pi / 180], rtol=1e-15)
    assert m.jnt_limited[0] == 1
    m = compile_xml(body.replace("30 70", "0.5 1.2"), head='<compiler angle="radian"/>')
    assert np.allclose(m.jnt_range[0], [0.5, 1.2], rtol=0)
    # a slide joint's range is a length in either case
    m = compile_xml('<body pos="0 0 1"><joint type="slide" axis="0 0 1" limited="true" range="-0.3 0.4"/>'
                    '<geom type="sphere" size="0.1"/></body>')
    assert np.allclose(m.jnt_range[0], [-0.3, 0.4], rtol=0)


def test_euler_is_intrinsic_xyz_in_degrees():
    m = compile_xml('<body pos="0 0 1" euler="90 0 0"><freejoint/><geom type="sphere" size="0.1"/></body>'
                    '<body pos="1 0 1" euler="90 180 0"><freejoint/><geom type="sphere" size="0.1"/></body>')
    s = math.sqrt(0.5)
    assert np.allclose(m.body_quat[1], [s, s, 0, 0], atol=1e-15)
    rx = np.array([[1, 0, 0], [0, 0, -1], [0, 1, 0]])
    ry = np.diag([-1.0, 1.0, -1.0])
    assert np.allclose(mjcf.quat_to_mat(m.body_quat[2]), rx @ ry, atol=1e-15)     # x first, then y about the NEW y axis
    # a free joint's reference configuration is the body's frame
    assert np.allclose(m.qpos0[:7], [0, 0, 1, s, s, 0, 0], atol=1e-15)
    assert np.allclose(m.qpos0[7:10], [1, 0, 1])


def test_nested_default_classes():
    head = ('<default><joint armature="1" damping="2" limited="true"/><geom density="5"/>'
            '<default class="leg"><joint damping="3"/>'
            '<default class="foot"><joint armature="0.5"/></default></default></default>')
    body = ('<body pos="0 0 1"><joint name="a" type="hinge" range="-10 10"/><geom type="sphere" size="0.1"/>'
            '<body pos="0 0 -0.2" childclass="leg"><joint name="b" type="hinge" range="-10 10"/>'
            '<geom type="sphere" size="0.1"/>'
            '<body pos="0 0 -0.2"><joint name="c" class="foot" type="hinge" range="-10 10"/>'
            '<joint name="d" type="hinge" axis="1 0 0" range="-10 10" damping="7"/>'
            '<geom type="sphere" size="0.1" density="9"/></body></body></body>')
    m = compile_xml(body, head=head)
    assert list(m.dof_armature) == [1.0, 1.0, 0.5, 1.0]                 # main, leg (inherits main), foot, leg
    assert list(m.dof_damping) == [2.0, 3.0, 3.0, 7.0]                  # main, leg, foot (inherits leg), own attribute
    assert list(m.jnt_limited) == [1, 1, 1, 1]
    mass = lambda rho: rho * 4 / 3 * math.pi * 1e-3
    assert np.allclose(m.body_mass[1:], [mass(5), mass(5), mass(9)], rtol=1e-14)


def test_body_numbering_is_depth_first_and_parent_child_pairs_do_not_collide():
    m = compile_xml('<geom name="floor" type="plane" size="5 5 1"/>'
                    '<body name="a" pos="0 0 1"><freejoint/><geom name="ga" type="sphere" size="0.2"/>'
                    '<body name="a1" pos="0.3 0 0"><joint type="hinge"/><geom name="ga1" type="sphere" size="0.2"/>'
                    '<body name="a2" pos="0.3 0 0"><joint type="hinge"/><geom name="ga2" type="sphere" size="0.2"/></body>'
                    '</body></body>'
                    '<body name="b" pos="2 0 1"><freejoint/><geom name="gb" type="sphere" size="0.2"/></body>')
    assert m.names["body"] == ["world", "a", "a1", "a2", "b"]
    assert list(m.body_parentid) == [0, 0, 1, 2, 0]
    pairs = {tuple(sorted((m.names["geom"][a], m.names["geom"][b]))) for a, b in m.pair_geom if a >= 0}
    # every geom against the floor; a-a2 (grandparent) and everything against b; NOT a-a1, a1-a2 (parent and child)
    assert pairs == {("floor", "ga"), ("floor", "ga1"), ("floor", "ga2"), ("floor", "gb"), ("ga", "ga2"),
                     ("ga", "gb"), ("ga1", "gb"), ("ga2", "gb")}


# ---------------------------------------------------------------------------------------- constants at qpos0
def test_invweight0_of_one_free_body():
    a, b, c, rho = 0.3, 0.2, 0.1, 40.0
    m = compile_xml(f'<body pos="0 0 1"><freejoint/><geom type="box" size="{a} {b} {c}" density="{rho}"/></body>')
    mass = rho * 8 * a * b * c
    inertia = mass / 3 * np.array([b * b + c * c, a * a + c * c, a * a + b * b])
    assert np.allclose(m.body_invweight0[1], [1 / mass, np.mean(1 / inertia)], rtol=1e-13)
    assert np.allclose(m.dof_invweight0, [1 / mass] * 3 + [np.mean(1 / inertia)] * 3, rtol=1e-13)
    assert m.meaninertia == pytest.approx((3 * mass + inertia.sum()) / 6, rel=1e-13)
    assert np.allclose(m.body_invweight0[0], 0)


def test_invweight0_of_a_hinge_pendulum():
    r, length, rho, arm = 0.05, 0.7, 1000.0, 0.01
    m = compile_xml(f'<body pos="0 0 2"><joint type="hinge" axis="0 1 0" armature="{arm}"/>'
                    f'<geom type="sphere" size="{r}" pos="0 0 {-length}"/></body>')
    mass = rho * 4 / 3 * math.pi * r ** 3
    about_axis = 0.4 * mass * r * r + mass * length * length + arm
    assert m.dof_invweight0[0] == pytest.approx(1 / about_axis, rel=1e-13)
    assert m.meaninertia == pytest.approx(about_axis, rel=1e-13)
    # the bob's centre moves on a circle: J_trans = axis x arm (length l), J_rot = axis; mean of the diagonal of J M^-1 J'
    assert np.allclose(m.body_invweight0[1], [length ** 2 / about_axis / 3, 1 / about_axis / 3], rtol=1e-13)


def test_invweight0_free_dofs_average_three_and_three():
    # a free body carrying a hinged arm: the free joint's translational dofs share one value, its rotational dofs another
    m = compile_xml('<body pos="0 0 1"><freejoint/><geom type="box" size="0.3 0.2 0.1"/>'
                    '<body pos="0.4 0 0"><joint type="hinge" axis="0 1 0"/><geom type="capsule" size="0.05 0.2"/></body></body>')
    mass_matrix, _, _ = mjcf.mass_matrix_numpy(m, m.qpos0)
    diag = np.diag(np.linalg.inv(mass_matrix))
    assert np.allclose(m.dof_invweight0[:3], diag[:3].mean(), rtol=1e-13)
    assert np.allclose(m.dof_invweight0[3:6], diag[3:6].mean(), rtol=1e-13)
    assert m.dof_invweight0[6] == pytest.approx(diag[6], rel=1e-13)
    assert len(set(np.round(diag[3:6] / diag[3:6].mean(), 6))) > 1     # (the three really differ before averaging)
    # the mass matrix itself: total mass on the translational diagonal
    assert np.allclose(np.diag(mass_matrix)[:3], m.body_mass.sum(), rtol=1e-13)


def test_the_ant_levels_total_mass_at_density_5():
    """The shipped 2-agent level, by hand: per ant a torso sphere r 0.25, four legs of three capsules r 0.08 (the XML's
    fromto lengths), one camera-less body each -- summed with the closed forms above at the level's density 5."""
    from mjrl_amd import levels
    m = mjcf.compile_mjcf(levels.level_path("two_agent.xml"))
    rho = 5.0
    total = 0.0
    for g in range(m.ngeom):
        if m.geom_bodyid[g] == 0 or m.body_weldid[m.geom_bodyid[g]] == 0:
            continue
        t, s = m.geom_type[g], m.geom_size[g]
        if t == mjcf.GEOM_SPHERE:
            total += rho * 4 / 3 * math.pi * s[0] ** 3
        elif t == mjcf.GEOM_CAPSULE:
            total += rho * (math.pi * s[0] ** 2 * 2 * s[1] + 4 / 3 * math.pi * s[0] ** 3)
        elif t == mjcf.GEOM_BOX:
            total += rho * 8 * s[0] * s[1] * s[2]
    moving = [b for b in range(1, m.nbody) if m.body_weldid[b] != 0]
    assert m.body_mass[moving].sum() == pytest.approx(total, rel=1e-13)
    torso = m.name2id("body", "sender")
    assert m.body_mass[torso] == pytest.approx(rho * 4 / 3 * math.pi * 0.25 ** 3, rel=1e-13)


# ---------------------------------------------------------------------------------------- <inertial>, <exclude>, <keyframe>
def test_inertial_elements_override_the_geoms_unless_the_compiler_says_otherwise():
    body = ('<body pos="0 0 1"><freejoint/><inertial pos="0.1 0 0.2" mass="3" diaginertia="0.3 0.2 0.1" euler="0 0 90"/>'
            '<geom type="sphere" size="0.1"/></body>')
    m = compile_xml(body)                                               # inertiafromgeom="auto": the <inertial> wins
    assert m.body_mass[1] == 3.0 and np.allclose(m.body_ipos[1], [0.1, 0, 0.2]) and np.allclose(m.body_inertia[1], [0.3, 0.2, 0.1])
    assert np.allclose(m.body_iquat[1], [math.sqrt(0.5), 0, 0, math.sqrt(0.5)])
    assert np.allclose(m.body_invweight0[1], [1 / 3.0, np.mean(1 / np.array([0.3, 0.2, 0.1]))], rtol=1e-12)
    m = compile_xml(body, head='<compiler inertiafromgeom="true"/>')    # the levels' setting: geoms only
    assert m.body_mass[1] == pytest.approx(1000 * 4 / 3 * math.pi * 1e-3, rel=1e-13) and np.allclose(m.body_ipos[1], 0)
    m = compile_xml(body, head='<compiler inertiafromgeom="false"/>')
    assert m.body_mass[1] == 3.0
    with pytest.raises(ValueError, match="no mass"):                    # "false" and no <inertial> on a body that moves
        compile_xml('<body pos="0 0 1"><freejoint/><geom type="sphere" size="0.1"/></body>',
                    head='<compiler inertiafromgeom="false"/>')


def test_fullinertia_is_brought_to_its_principal_frame():
    # the tensor of a box (principal moments 0.5, 0.3, 0.2) turned by 30 degrees about z, given in the body frame
    cz, sz = math.cos(math.radians(30)), math.sin(math.radians(30))
    rot = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    t = rot @ np.diag([0.5, 0.3, 0.2]) @ rot.T
    full = f"{t[0, 0]} {t[1, 1]} {t[2, 2]} {t[0, 1]} {t[0, 2]} {t[1, 2]}"
    m = compile_xml(f'<body pos="0 0 1"><freejoint/><inertial pos="0 0 0" mass="2" fullinertia="{full}"/>'
                    '<geom type="sphere" size="0.1"/></body>')
    assert np.allclose(m.body_inertia[1], [0.5, 0.3, 0.2], rtol=1e-12)
    assert np.allclose(inertia_tensor(m, 1), t, rtol=1e-12, atol=1e-15)
    assert np.linalg.det(mjcf.quat_to_mat(m.body_iquat[1])) == pytest.approx(1.0, abs=1e-13)


def test_contact_exclude_removes_the_two_bodies_pairs_and_keyframes_change_nothing():
    plain = compile_xml(TWO)
    assert {tuple(sorted((plain.names["geom"][a], plain.names["geom"][b]))) for a, b in plain.pair_geom if a >= 0} == {("ga", "gb")}
    m = compile_xml(TWO, tail='<contact><exclude body1="a" body2="b"/></contact>')
    assert not [1 for a, b in m.pair_geom if a >= 0]
    with pytest.raises(KeyError):
        compile_xml(TWO, tail='<contact><exclude body1="a" body2="nobody"/></contact>')
    from mjrl_amd import blob
    keyed = compile_xml(TWO, tail='<keyframe><key qpos="0.3 0.4"/></keyframe>')      # mj_resetData goes to qpos0, not to a key
    assert blob.pack(keyed) == blob.pack(plain) and np.array_equal(keyed.qpos0, [0, 0])


# ---------------------------------------------------------------------------------------- the subset is loud
BALL = '<body pos="0 0 1"><freejoint/><geom name="g" type="sphere" size="0.1"/></body>'
TWO = ('<body name="a" pos="0 0 1"><joint name="ja" type="hinge"/><geom name="ga" type="sphere" size="0.1"/></body>'
       '<body name="b" pos="1 0 1"><joint name="jb" type="hinge"/><geom name="gb" type="sphere" size="0.1"/></body>')


@pytest.mark.parametrize("body, head, tail, named", [
    (TWO, "", '<equality><weld body1="a" body2="b"/></equality>', "equality"),
    (TWO, "", '<tendon><fixed><joint joint="ja" coef="1"/></fixed></tendon>', "tendon"),
    (TWO, "", '<contact><pair geom1="ga" geom2="gb" condim="1"/></contact>', "pair"),
    (BALL, '<include file="other.xml"/>', "", "include"),
    (BALL, '<option solver="Newton"/>', "", "solver"),
    (BALL, '<option solver="CG"/>', "", "solver"),
    (BALL, '<option cone="elliptic"/>', "", "cone"),
    (BALL, '<option noslip_iterations="3"/>', "", "noslip_iterations"),
    (BALL, '<option integrator="implicit"/>', "", "integrator"),
    (BALL, '<option viscosity="0.1"/>', "", "viscosity"),
    (BALL, '<option><flag contact="disable"/></option>', "", "flag"),
    (BALL, '<compiler coordinate="global"/>', "", "coordinate"),
    (BALL, '<compiler autolimits="true"/>', "", "autolimits"),
    (BALL, '<asset><mesh name="m" file="m.stl"/></asset>', "", "mesh"),
    (BALL, '<default><tendon width="0.1"/></default>', "", "tendon"),
    ('<body pos="0 0 1"><joint type="ball"/><geom type="sphere" size="0.1"/></body>', "", "", "ball"),
    ('<body pos="0 0 1"><joint type="hinge" frictionloss="0.1"/><geom type="sphere" size="0.1"/></body>', "", "",
     "frictionloss"),
    ('<body pos="0 0 1"><joint type="free" stiffness="2"/><geom type="sphere" size="0.1"/></body>', "", "", "stiffness"),
    ('<body pos="0 0 1"><joint type="hinge" springdamper="1 1"/><geom type="sphere" size="0.1"/></body>', "", "", "springdamper"),
    ('<body pos="0 0 1"><freejoint/><geom type="ellipsoid" size="0.1 0.2 0.3"/></body>', "", "", "ellipsoid"),
    ('<body pos="0 0 1"><freejoint/><geom type="cylinder" size="0.1 0.2"/></body>', "", "", "cylinder"),
    ('<body pos="0 0 1" mocap="true"><geom type="sphere" size="0.1"/></body>', "", "", "mocap"),
    ('<body pos="0 0 1" gravcomp="1"><freejoint/><geom type="sphere" size="0.1"/></body>', "", "", "gravcomp"),
    ('<body pos="0 0 1"><freejoint/><geom type="sphere" size="0.1" priority="2"/></body>', "", "", "priority"),
    (TWO, "", '<actuator><position joint="ja" kp="10"/></actuator>', "position"),
    (TWO, "", '<actuator><general joint="ja" gainprm="3"/></actuator>', "gainprm"),
    (TWO, "", '<actuator><motor joint="ja" forcelimited="true" forcerange="-1 1"/></actuator>', "forcelimited"),
    ('<body pos="0 0 1"><freejoint/><geom type="sphere" size="0.1"/><site name="s"/></body>', "",
     '<sensor><gyro site="s"/></sensor>', "gyro"),
    ('<body pos="0 0 1"><freejoint/><geom type="sphere" size="0.1"/><site name="s" type="box" size="0.1 0.1 0.1"/></body>', "",
     '<sensor><touch site="s"/></sensor>', "touch sensor on a site of type"),
    ('<body pos="0 0 1"><freejoint/><geom type="sphere" size="0.1"/><light mode="trackcom" pos="0 0 2"/></body>', "", "",
     "light mode"),
    ('<body pos="0 0 1"><freejoint/><geom type="sphere" size="0.1" group="3"/><camera name="c"/></body>', "", "", "group 3"),
    ('<body pos="0 0 1"><freejoint/><geom type="sphere" size="0.1" condim="4"/></body>', "", "", "condim 4"),
    ('<body pos="0 0 1"><freejoint/><geom type="sphere" size="0.1" condim="6"/></body>', "", "", "condim 6"),
    ('<body pos="0 0 1"><freejoint/><geom type="sphere" size="0.1" solref="-1000 -10"/></body>', "", "", "solref"),
    ('<body pos="0 0 1"><joint type="hinge" limited="true" range="-1 1" solreflimit="-100 -1"/><geom type="sphere" size="0.1"/></body>',
     "", "", "solreflimit"),
])
def test_features_outside_the_subset_are_refused_by_name(body, head, tail, named):
    with pytest.raises(mjcf.UnsupportedMJCF) as err:
        compile_xml(body, head=head, tail=tail)
    assert named in str(err.value)
    assert isinstance(err.value, ValueError)


@pytest.mark.parametrize("head", [
    '<option solver="PGS" cone="pyramidal" noslip_iterations="0" timestep="0.01" integrator="RK4"/>',
    '<option viscosity="0" wind="0 0 0"><flag energy="disable" contact="enable"/></option>',
    '<compiler angle="radian" coordinate="local" inertiafromgeom="true"/>',
    '<size nconmax="50" njmax="200"/><custom><numeric name="n" data="1 2"/></custom>',
    '<asset><texture name="t" type="2d" builtin="checker" width="8" height="8"/>'
    '<material name="m" texture="t" reflectance="0.5"/></asset>',
])
def test_what_is_implemented_or_inert_still_compiles(head):
    body = ('<body pos="0 0 1"><joint type="hinge" stiffness="0" frictionloss="0"/>'
            '<geom type="sphere" size="0.1" priority="0"/></body>')
    assert compile_xml(body, head=head).nv == 1


def test_statistic_meaninertia_overrides_the_computed_value():
    """<statistic meaninertia> replaces the mean diagonal of M(qpos0) in the model (the scale of the solver's stop test);
    the section's other attributes are the visualiser's."""
    body = '<body pos="0 0 1"><freejoint/><geom type="sphere" size="0.1" density="1000"/></body>'
    computed = compile_xml(body).meaninertia
    assert abs(computed - np.mean([4.18879, 4.18879, 4.18879, 0.0167552, 0.0167552, 0.0167552])) < 1e-4
    assert compile_xml(body, head='<statistic meaninertia="2.5" extent="3"/>').meaninertia == 2.5
    assert compile_xml(body, head='<statistic extent="3"/>').meaninertia == computed


def test_every_shipped_level_is_inside_the_subset():
    from mjrl_amd import levels
    for name in sorted(levels.LEVELS):
        assert mjcf.compile_mjcf(levels.level_path(name)).nbody > 1
