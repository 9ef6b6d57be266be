"""Known-answer tests that pin the CPU oracle (physics parity is otherwise unpinned: the reference
asserts no physics value anywhere and mujoco itself cannot be run here -- SURVEY.md section 8c).

Each case has a closed-form answer: discrete free fall, conservation laws of an unforced system,
the mass matrix from an independent Jacobian-sum formula, static contact force = weight, the
rangefinder distance to a known wall.
"""
import numpy as np
import pytest

from mjrl_amd import blob, levels, mjcf
from oracle.oracle import OracleEnv


def make(xml: str, **kw):
    model = mjcf.compile_mjcf_string(xml, **kw)
    return model, OracleEnv(blob.pack(model))


FREE_SPHERE = """
<mujoco><option timestep="0.002"/><worldbody>
  <body name="ball" pos="0 0 5"><joint type="free" name="root"/>
    <geom type="sphere" size="0.1" density="1000"/></body>
</worldbody></mujoco>"""


def test_free_fall_matches_discrete_closed_form():
    model, env = make(FREE_SPHERE)
    h, g, n = model.timestep, 9.81, 500
    env.step(n)
    # semi-implicit Euler: v_k = -g h k ; z_n = z_0 - g h^2 n (n+1) / 2
    assert env.qvel[2] == pytest.approx(-g * h * n, rel=1e-12)
    assert env.qpos[2] == pytest.approx(5.0 - g * h * h * n * (n + 1) / 2, rel=1e-12)
    assert env.ncon == 0 and env.nefc == 0


TUMBLER = """
<mujoco><option timestep="0.0005" gravity="0 0 0"/><worldbody>
  <body name="brick" pos="0 0 1"><joint type="free" name="root"/>
    <geom type="box" size="0.1 0.2 0.3" density="800" contype="0" conaffinity="0"/></body>
</worldbody></mujoco>"""


def test_torque_free_body_conserves_angular_momentum_and_energy():
    model, env = make(TUMBLER)
    env.qvel[:] = [0.3, -0.2, 0.1, 2.0, 0.5, -1.0]
    inertia = model.body_inertia[1]

    def momentum_energy():
        env.forward()
        rot = env.ximat[1].reshape(3, 3)
        # free-joint angular velocity is expressed in the body frame (iquat is identity for one box)
        w_local = env.qvel[3:6]
        l_world = rot @ (inertia * (rot.T @ (env.xmat[1].reshape(3, 3) @ w_local)))
        kinetic = 0.5 * env.qvel @ env.qMdense @ env.qvel
        return l_world, kinetic

    l0, e0 = momentum_energy()
    env.step(2000)
    l1, e1 = momentum_energy()
    assert np.allclose(l0, l1, rtol=2e-3, atol=1e-6)
    assert e1 == pytest.approx(e0, rel=2e-3)
    # linear motion of the centre of mass is uniform
    assert np.allclose(env.qpos[:3], np.array([0, 0, 1]) + np.array([0.3, -0.2, 0.1]) * env.time, atol=1e-9)


DOUBLE_PENDULUM = """
<mujoco><option timestep="0.0002"/><worldbody>
  <body name="upper" pos="0 0 2"><joint type="hinge" axis="0 1 0" name="j1"/>
    <geom type="capsule" fromto="0 0 0 0 0 -0.5" size="0.05" density="500" contype="0" conaffinity="0"/>
    <body name="lower" pos="0 0 -0.5"><joint type="hinge" axis="0.3 1 0" name="j2"/>
      <geom type="capsule" fromto="0 0 0 0.1 0 -0.4" size="0.04" density="700" contype="0" conaffinity="0"/>
    </body></body>
</worldbody></mujoco>"""


def _energy(model, env):
    env.forward()
    kinetic = 0.5 * env.qvel @ env.qMdense @ env.qvel
    potential = sum(model.body_mass[b] * 9.81 * env.xipos[b, 2] for b in range(1, model.nbody))
    return kinetic + potential


def test_double_pendulum_energy_is_conserved():
    """Exercises the Coriolis/centrifugal part of the RNE bias and the sparse factorisation."""
    model, env = make(DOUBLE_PENDULUM)
    env.qpos[:] = [1.0, -0.7]
    env.qvel[:] = [0.5, 2.0]
    e0 = _energy(model, env)
    env.step(5000)   # one second
    e1 = _energy(model, env)
    assert abs(e1 - e0) < 2e-3 * abs(e0)
    assert abs(env.qpos[0] - 1.0) > 0.1   # it did move


def test_mass_matrix_matches_jacobian_sum_on_the_two_agent_level():
    model = mjcf.compile_mjcf(levels.level_path("two_agent.xml"))
    env = OracleEnv(blob.pack(model))
    rng = np.random.default_rng(7)
    for _ in range(3):
        env.qpos[:] = model.qpos0 + 0.4 * rng.normal(size=model.nq)
        env.forward()
        expect, _, _ = mjcf.mass_matrix_numpy(model, env.qpos.copy())
        assert np.allclose(env.qMdense, expect, rtol=0, atol=1e-13)


def test_gravity_bias_matches_potential_gradient():
    model = mjcf.compile_mjcf(levels.level_path("two_agent.xml"))
    env = OracleEnv(blob.pack(model))
    rng = np.random.default_rng(3)
    env.qpos[:] = model.qpos0 + 0.3 * rng.normal(size=model.nq)
    env.qvel[:] = 0
    env.forward()
    q = env.qpos.copy()
    xpos, xquat = mjcf.kinematics_numpy(model, q)
    expect = np.zeros(model.nv)
    for b in range(1, model.nbody):
        if model.body_lastdof[b] < 0:
            continue
        com = xpos[b] + mjcf.quat_to_mat(xquat[b]) @ model.body_ipos[b]
        jac = mjcf.body_jacobian_numpy(model, xpos, xquat, b, com)
        expect += jac[:3].T @ (model.body_mass[b] * np.array([0, 0, 9.81]))
    assert np.allclose(env.qfrc_bias, expect, atol=1e-12)


RESTING_BALL = """
<mujoco><option timestep="0.002"/><worldbody>
  <geom type="plane" size="5 5 0.1"/>
  <body name="ball" pos="0 0 0.1"><joint type="free" name="root"/>
    <geom type="sphere" size="0.1" density="1000"/></body>
</worldbody></mujoco>"""


def test_resting_contact_force_equals_weight():
    model, env = make(RESTING_BALL)
    env.step(1500)
    assert env.ncon == 1
    con = env.contacts()[0]
    weight = model.body_mass[1] * 9.81
    assert con["normal_force"] == pytest.approx(weight, rel=1e-4)
    assert abs(env.qvel[2]) < 1e-6
    assert con["dist"] < 0            # soft contact: small penetration
    assert con["dist"] > -0.01
    assert np.allclose(con["frame"][0], [0, 0, 1])


SLIDING_BOX = """
<mujoco><option timestep="0.002"/><worldbody>
  <geom type="plane" size="50 50 0.1" friction="0.5 0.005 0.0001"/>
  <body name="puck" pos="0 0 0.1"><joint type="free" name="root"/>
    <geom type="sphere" size="0.1" density="1000" friction="0.5 0.005 0.0001"/></body>
</worldbody></mujoco>"""


def test_friction_pyramid_decelerates_a_sliding_ball():
    """A ball sliding (not yet rolling) on the plane loses linear momentum through friction and picks up spin."""
    model, env = make(SLIDING_BOX)
    env.step(300)                      # settle
    env.qvel[0] = 1.0
    env.step(50)
    assert 0.0 < env.qvel[0] < 1.0     # friction acts against sliding
    assert env.qvel[4] > 0.0           # and spins the ball up about +y (local frame still ~ world)


def test_joint_limits_of_the_ant_are_active_at_reset():
    """ankle ranges exclude qpos0, so eight limit rows exist from the first step."""
    model = mjcf.compile_mjcf(levels.level_path("two_agent.xml"))
    env = OracleEnv(blob.pack(model))
    assert env.ncon == 0 and env.nefc == 8
    assert np.allclose(np.abs(env.efc_pos[:8]), np.radians(30), atol=1e-12)
    assert np.all(env.efc_pos[:8] < 0)
    assert np.all(env.efc_force[:8] > 0)


def test_rangefinder_measures_the_wall_distance():
    """Testing/sensor_levels/Model3.xml geometry: box at x=4.595 flipped 180 deg about x; its site looks along
    the site z axis (world -z after the flip) from 0.474 above the floor."""
    model = mjcf.compile_mjcf(levels.level_path("sensor_rangefinder.xml"))
    env = OracleEnv(blob.pack(model))
    assert np.allclose(env.site_xmat[0].reshape(3, 3)[:, 2], [0, 0, -1], atol=1e-12)
    assert env.sensordata[0] == pytest.approx(0.4743838, abs=1e-9)
    # aim it at the +y wall instead: border2's inner face is at y = 4.738263 - 0.25
    half = np.sqrt(0.5)
    env.qpos[3:7] = [half, -half, 0, 0]     # rotate -90 deg about x: local z -> world +y, local x unchanged
    env.qpos[2] = 0.6
    env.forward()
    assert np.allclose(env.site_xmat[0].reshape(3, 3)[:, 2], [0, 1, 0], atol=1e-12)
    assert env.sensordata[0] == pytest.approx(4.738263 - 0.25 - env.site_xpos[0, 1], abs=1e-9)


def test_frame_axis_sensor_reports_site_x_axis():
    model = mjcf.compile_mjcf(levels.level_path("sensor_framexaxis.xml"))
    env = OracleEnv(blob.pack(model))
    assert np.allclose(env.sensordata[:3], [1, 0, 0], atol=1e-12)


def test_accelerometer_reads_zero_in_free_fall_and_g_at_rest():
    xml = """
    <mujoco><worldbody><geom type="plane" size="5 5 0.1"/>
      <body name="ball" pos="0 0 1"><joint type="free" name="root"/>
        <geom type="sphere" size="0.1" density="1000"/><site name="s" pos="0 0 0" size="0.01"/></body>
    </worldbody><sensor><accelerometer name="acc" site="s" cutoff="50"/></sensor></mujoco>"""
    model, env = make(xml)
    env.step(10)
    assert np.allclose(env.sensordata[:3], 0, atol=1e-9)           # free fall: proper acceleration is zero
    env.qpos[2] = 0.1
    env.qvel[:] = 0
    env.step(1500)
    assert np.allclose(env.sensordata[:3], [0, 0, 9.81], atol=1e-3)  # at rest the floor pushes up with g


def _incline(theta_deg, mu):
    """A box on a plane with gravity tilted by theta about y (the same thing as a slope of angle theta)."""
    th = np.deg2rad(theta_deg)
    return f"""
<mujoco><option timestep="0.002" gravity="{9.81 * np.sin(th)} 0 {-9.81 * np.cos(th)}"/><worldbody>
  <geom type="plane" size="50 50 0.1" friction="{mu} 0.005 0.0001"/>
  <body name="block" pos="0 0 0.1"><joint type="free" name="root"/>
    <geom type="box" size="0.2 0.2 0.1" density="1000" friction="{mu} 0.005 0.0001"/></body>
</worldbody></mujoco>"""


def test_coulomb_friction_holds_below_the_friction_angle_and_slides_above_it():
    """Known answer of dry friction (pyramidal cone, slope along a pyramid axis): below tan(theta) = mu the block stays
    (a soft-constraint solver lets it creep, orders of magnitude slower than free sliding); above it the block
    accelerates with g (sin(theta) - mu cos(theta))."""
    mu = 0.5
    model, env = make(_incline(20.0, mu))                 # tan 20 deg = 0.364 < 0.5
    env.step(1500)
    assert env.ncon == 4
    v_hold = env.qvel[0]
    free_fall_speed = 9.81 * np.sin(np.deg2rad(20.0)) * 3.0
    assert abs(v_hold) < 2e-3 * free_fall_speed
    theta = 35.0                                          # tan 35 deg = 0.700 > 0.5
    model, env = make(_incline(theta, mu))
    env.step(500)                                         # settle onto the plane while sliding
    v0 = env.qvel[0]
    normal = []
    for _ in range(500):
        env.step()
        normal.append(sum(c["normal_force"] for c in env.contacts()))
    accel = (env.qvel[0] - v0) / (500 * 0.002)
    expect = 9.81 * (np.sin(np.deg2rad(theta)) - mu * np.cos(np.deg2rad(theta)))
    assert accel == pytest.approx(expect, rel=2e-2)
    # (the sliding block chatters on its leading edge; on average the plane carries the normal component of its weight)
    assert np.mean(normal) == pytest.approx(model.body_mass[1] * 9.81 * np.cos(np.deg2rad(theta)), rel=5e-2)


def test_soft_contact_rests_at_the_depth_its_reference_acceleration_dictates():
    """MuJoCo's soft-constraint model in closed form for one frictionless normal row at rest: the reference acceleration
    aref = -k d (pos - margin) - b v with k = 1 / (dmax^2 tc^2 dr^2), and force f = (aref - a0) / (A + R) with a0 = -g the
    unconstrained acceleration along the normal, A = 1 / m, R = (1 - d) / d * diagApprox.  At rest f = m g, so the
    penetration follows from the numbers in the model (solref 0.02 1, solimp 0.9 0.95 0.001 0.5 2)."""
    xml = """
<mujoco><option timestep="0.002"/><worldbody>
  <geom type="plane" size="5 5 0.1" condim="1"/>
  <body name="ball" pos="0 0 0.1"><joint type="free" name="root"/><geom type="sphere" size="0.1" density="1000" condim="1"/></body>
</worldbody></mujoco>"""
    model, env = make(xml)
    env.step(3000)
    assert env.ncon == 1 and env.nefc == 1 and abs(env.qvel[2]) < 1e-7
    pen = -env.contacts()[0]["dist"]
    m, g = model.body_mass[1], 9.81
    dmin, dmax, width, mid, power = 0.9, 0.95, 0.001, 0.5, 2.0
    tc, dr = 0.02, 1.0
    k = 1.0 / (dmax * dmax * tc * tc * dr * dr)

    def imp(r):
        x = min(abs(r) / width, 1.0)
        y = (x / mid) ** power * mid if x <= mid else 1 - ((1 - x) / (1 - mid)) ** power * (1 - mid)
        return dmin + y * (dmax - dmin)
    # rest: qacc = 0 = a0 + f / m  and  (A + R) f = aref - a0  with  R = (1 - d) / d * A  (diagApprox = 1 / m for the free body)
    # => f = m g  and  aref = a0 + (A + R) m g = -g + g / d  =>  k d pen = g (1 / d - 1)
    lo, hi = 0.0, 0.01
    for _ in range(200):
        r = 0.5 * (lo + hi)
        d = imp(r)
        if k * d * r < g * (1.0 / d - 1.0): lo = r
        else: hi = r
    assert pen == pytest.approx(0.5 * (lo + hi), rel=1e-3)


def _contact_state(level, steps=260, seed=11, tight=True):
    """A contact-rich state of an ant level in the oracle (random actions from just above the floor); with `tight` the
    solver runs to its fixed point (tolerance 0, 4000 sweeps)."""
    model = mjcf.compile_mjcf(levels.level_path(level))
    if tight:
        model.tolerance, model.iterations = 0.0, 4000
    env = OracleEnv(blob.pack(model))
    rng = np.random.default_rng(seed)
    for j in range(model.njnt):
        if model.jnt_type[j] == mjcf.JNT_FREE:
            env.qpos[model.jnt_qposadr[j] + 2] = 0.6
    for k in range(2000):                       # (the ants hop: wait for a step with several feet on the ground)
        env.ctrl[:] = rng.uniform(-1, 1, model.nu)
        env.step()
        if k >= steps and env.ncon >= 3:
            break
    env.ctrl[:] = rng.uniform(-1, 1, model.nu)
    env.forward()
    return model, env


@pytest.mark.parametrize("level", ["two_agent.xml", "four_agent.xml"])
def test_solver_output_satisfies_the_complementarity_conditions(level):
    """What the constraint solver returns is checked against the problem it is meant to solve, in numpy and without any
    of its code: with A = J M^-1 J' + diag(R) and b = J qacc_smooth - aref, the forces of the pyramidal-cone dual obey
    f >= 0, g = A f + b >= 0 and f'g = 0 (run to the fixed point: tolerance 0), qfrc_constraint is J'f and qacc is
    qacc_smooth + M^-1 J'f."""
    model, env = _contact_state(level)
    n = env.nefc
    assert env.ncon >= 3 and n >= 8
    J, R, f = env.efc_J[:n].copy(), env.efc_R[:n].copy(), env.efc_force[:n].copy()
    M = env.qMdense.copy()
    A = J @ np.linalg.solve(M, J.T) + np.diag(R)
    b = J @ env.qacc_smooth - env.efc_aref[:n]
    g = A @ f + b
    scale = np.abs(b).max()
    assert (f >= 0).all()
    assert g.min() > -1e-7 * scale, g.min() / scale
    assert np.abs(f * g).max() < 1e-7 * scale * max(f.max(), 1.0)
    assert (f > 0).sum() >= 3                                   # the ants stand on something
    assert np.allclose(env.qfrc_constraint, J.T @ f, rtol=0, atol=1e-9 * max(1.0, np.abs(J.T @ f).max()))
    assert np.allclose(env.qacc, env.qacc_smooth + np.linalg.solve(M, J.T @ f), rtol=0, atol=1e-8 * max(1.0, np.abs(env.qacc).max()))


def test_contact_rows_are_the_pyramid_edges_of_the_relative_contact_velocity():
    """Every contact row against the independent numpy Jacobian of its two bodies: row 2k / 2k+1 of a contact is
    n' (J2 - J1) +/- mu t_k' (J2 - J1) at the contact point (the pyramidal cone's edges), J_b the translational Jacobian of
    the point moving with body b (mjcf.body_jacobian_numpy)."""
    model, env = _contact_state("two_agent.xml", tight=False)
    xpos, xquat = env.xpos.copy(), env.xquat.copy()
    checked = 0
    for c in env.contacts():
        adr = c["efc_address"]
        if adr < 0:
            continue
        b1, b2 = int(model.geom_bodyid[c["geom1"]]), int(model.geom_bodyid[c["geom2"]])
        dj = mjcf.body_jacobian_numpy(model, xpos, xquat, b2, c["pos"])[:3] - mjcf.body_jacobian_numpy(model, xpos, xquat, b1, c["pos"])[:3]
        frame = np.asarray(c["frame"]).reshape(3, 3)
        mu = max(model.geom_friction[c["geom1"]][0], model.geom_friction[c["geom2"]][0])
        for k in range(2):
            for s, sign in enumerate((1.0, -1.0)):
                expect = frame[0] @ dj + sign * mu * (frame[1 + k] @ dj)
                assert np.allclose(env.efc_J[adr + 2 * k + s], expect, rtol=0, atol=1e-12), (adr, k, s)
                checked += 1
    assert checked >= 8


def test_contact_rows_reference_acceleration_and_regularisation_follow_the_documented_formulas():
    """aref, R and the impedance of every contact row recomputed in numpy from MuJoCo's documented solver-parameter
    formulas (solref = (timeconst, dampratio), solimp = (d0, dwidth, width, midpoint, power); "Solver parameters" of the
    MuJoCo documentation): d(x) the power-law sigmoid of x = |pos - margin| / width, K = 1 / (dmax^2 tc^2 dr^2),
    B = 2 / (dmax tc), aref = -B vel - K d (pos - margin), R = (1 - d) / d * diagApprox (pyramid edges: 2 mu^2 times that,
    diagApprox = (1 + mu^2) (invweight0[body1] + invweight0[body2]))."""
    model, env = _contact_state("two_agent.xml", tight=False)
    n = env.nefc
    vel = env.efc_J[:n] @ env.qvel
    checked = 0
    for c in env.contacts():
        adr = c["efc_address"]
        if adr < 0:
            continue
        g1, g2 = c["geom1"], c["geom2"]
        assert np.array_equal(model.geom_solref[g1], model.geom_solref[g2]) and np.array_equal(model.geom_solimp[g1], model.geom_solimp[g2])
        tc, dr = model.geom_solref[g1]
        d0, dw, width, mid, power = model.geom_solimp[g1]
        tc = max(tc, 2 * model.timestep)
        margin = max(model.geom_margin[g1], model.geom_margin[g2]) - max(model.geom_gap[g1], model.geom_gap[g2])
        x = min(abs(c["dist"] - margin) / width, 1.0)
        if x <= mid:
            y = x ** power / mid ** (power - 1)
        else:
            y = 1.0 - (1.0 - x) ** power / (1.0 - mid) ** (power - 1)
        imp = d0 + y * (dw - d0)
        dmax = min(max(dw, 1e-4), 0.9999)
        K, B = 1.0 / (dmax * dmax * tc * tc * dr * dr), 2.0 / (dmax * tc)
        mu = max(model.geom_friction[g1][0], model.geom_friction[g2][0])
        b1, b2 = int(model.geom_bodyid[g1]), int(model.geom_bodyid[g2])
        tran = model.body_invweight0[b1][0] + model.body_invweight0[b2][0]
        R = 2 * mu * mu * (1 - imp) / imp * (tran + mu * mu * tran)
        for r in range(adr, adr + 4):
            assert np.isclose(env.efc_aref[r], -B * vel[r] - K * imp * (c["dist"] - margin), rtol=1e-12, atol=1e-12), r
            assert np.isclose(env.efc_R[r], R, rtol=1e-12), r
            checked += 1
    assert checked >= 8


def test_a_step_is_the_semi_implicit_euler_update_of_its_own_forward_pass():
    """One step of the 2-agent level against numpy: with the forces of the step's forward pass, the new velocity is
    v + h (M + h diag(damping))^-1 (qfrc_smooth + qfrc_constraint) (MuJoCo's Euler integrator with joint damping treated
    implicitly), hinge positions move by h v', a free joint's position by h v' and its quaternion by the rotation
    h |w'| about w' (body frame), renormalised."""
    model, env = _contact_state("two_agent.xml", tight=False)
    h = model.timestep
    q0, v0 = env.qpos.copy(), env.qvel.copy()
    env.step()
    M = env.qMdense.copy()
    rhs = env.qfrc_smooth + env.qfrc_constraint
    v1 = v0 + h * np.linalg.solve(M + h * np.diag(model.dof_damping), rhs)
    assert np.allclose(env.qvel, v1, rtol=0, atol=1e-11 * max(1.0, np.abs(v1).max()))
    for j in range(model.njnt):
        qa, da = int(model.jnt_qposadr[j]), int(model.jnt_dofadr[j])
        if model.jnt_type[j] == mjcf.JNT_FREE:
            assert np.allclose(env.qpos[qa:qa + 3], q0[qa:qa + 3] + h * v1[da:da + 3], atol=1e-13)
            w = v1[da + 3:da + 6]
            ang = h * np.linalg.norm(w)
            axis = w / np.linalg.norm(w)
            dq = np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * axis])
            a = q0[qa + 3:qa + 7] / np.linalg.norm(q0[qa + 3:qa + 7])
            q = np.array([a[0] * dq[0] - a[1:] @ dq[1:], *(a[0] * dq[1:] + dq[0] * a[1:] + np.cross(a[1:], dq[1:]))])
            assert np.allclose(env.qpos[qa + 3:qa + 7], q / np.linalg.norm(q), atol=1e-13)
        else:
            assert np.isclose(env.qpos[qa], q0[qa] + h * v1[da], atol=1e-13)


# ---------------------------------------------------------------------------------------- joint springs (round 4)
def test_a_mass_on_a_slide_spring_settles_where_the_spring_carries_it():
    """body/joint stiffness + springref: the passive force -k (q - springref) on a slide's coordinate.  A damped mass on a
    vertical spring comes to rest at q = springref - m g / k."""
    k, ref = 400.0, 0.05
    model, env = make(f"""<mujoco><option timestep="0.001"/><worldbody>
      <body pos="0 0 1"><joint type="slide" axis="0 0 1" stiffness="{k}" springref="{ref}" damping="60"/>
        <geom type="sphere" size="0.1" density="1000" contype="0" conaffinity="0"/></body></worldbody></mujoco>""")
    mass = 1000 * 4 / 3 * np.pi * 1e-3
    env.step(8000)
    assert abs(env.qvel[0]) < 1e-9
    assert env.qpos[0] == pytest.approx(ref - mass * 9.81 / k, rel=1e-9)


def test_a_torsion_spring_swings_with_the_period_of_its_inertia():
    """No gravity, no damping: I q'' = -k q.  The semi-implicit Euler map x -> (q + h v', v' = v - h w^2 q) has the exact
    discrete frequency cos(w_d h) = 1 - (w h)^2 / 2; the zero crossings of q are half a discrete period apart."""
    k, length, r, arm, h = 3.0, 0.4, 0.05, 0.02, 0.001
    model, env = make(f"""<mujoco><option timestep="{h}" gravity="0 0 0"/><worldbody>
      <body pos="0 0 1"><joint type="hinge" axis="0 1 0" stiffness="{k}" armature="{arm}"/>
        <geom type="sphere" size="{r}" pos="0 0 {-length}" density="1000" contype="0" conaffinity="0"/></body></worldbody></mujoco>""")
    mass = 1000 * 4 / 3 * np.pi * r ** 3
    inertia = 0.4 * mass * r * r + mass * length * length + arm
    w = np.sqrt(k / inertia)
    period = 2 * np.pi / (np.arccos(1 - (w * h) ** 2 / 2) / h)
    env.qpos[0] = 0.3
    crossings, prev = [], env.qpos[0]
    for step in range(1, int(3.2 * period / h)):
        env.step()
        q = env.qpos[0]
        if prev > 0 >= q or prev < 0 <= q:
            crossings.append((step - 1 + prev / (prev - q)) * h)        # linear interpolation of the crossing
        prev = q
    assert len(crossings) >= 6
    half_periods = np.diff(crossings)
    assert np.allclose(half_periods, period / 2, rtol=2e-4)
    assert abs(period - 2 * np.pi / w) / period < 1e-3                 # (and that is the continuous period to O(h^2))


def test_springref_and_ref_of_a_hinge_are_angles_in_the_compilers_unit():
    xml = """<mujoco><option timestep="0.002" gravity="0 0 0"/><worldbody>
      <body pos="0 0 1"><joint type="hinge" axis="0 1 0" stiffness="5" springref="30" ref="10" damping="0.4"/>
        <geom type="sphere" size="0.05" pos="0 0 -0.3" density="1000" contype="0" conaffinity="0"/></body></worldbody></mujoco>"""
    model, env = make(xml)
    assert model.qpos0[0] == pytest.approx(np.radians(10)) and model.dof_springref[0] == pytest.approx(np.radians(30))
    assert model.dof_stiffness[0] == 5.0 and model.dof_qposadr[0] == 0
    env.step(5000)
    assert env.qpos[0] == pytest.approx(np.radians(30), abs=1e-9) and abs(env.qvel[0]) < 1e-9
    radian, _ = make(xml.replace("<option", '<compiler angle="radian"/><option').replace('springref="30" ref="10"', 'springref="0.5" ref="0.2"'))
    assert radian.qpos0[0] == 0.2 and radian.dof_springref[0] == 0.5
    with pytest.raises(mjcf.UnsupportedMJCF, match="free joint"):
        make('<mujoco><worldbody><body pos="0 0 1"><joint type="free" stiffness="2"/><geom type="sphere" size="0.1"/></body>'
             '</worldbody></mujoco>')
