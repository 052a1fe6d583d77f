"""Crazyflie airframe constants, derived from the link geometry on the host (init-time only).

Restates what the reference evaluates once per env construction and what the HIP kernels take as
constants:
  quad_models.py:1-42 (crazyflie_params) -> inertia.py:182-309 (QuadLink: box/cylinder links, parallel-axis
  theorem, centre-of-mass shift) -> quadrotor_dynamics.py:104-166 (update_model: thrust_max, torque_max,
  prop cross-products, arm) with the dynamics_change of swarm_rl/env_wrappers/quad_utils.py:31.
"""
import math

import numpy as np

GRAV = 9.81


def _box_inertia(l, w, h, m):  # inertia.py:88-96 (BoxLink.I_com)
    return np.diag([m * (h ** 2 + w ** 2) / 12.0, m * (l ** 2 + h ** 2) / 12.0, m * (w ** 2 + l ** 2) / 12.0])


def _cyl_inertia(h, r, m):  # inertia.py:147-154 (CylinderLink.I_com)
    return np.diag([m * (3 * r ** 2 + h ** 2) / 12.0, m * (3 * r ** 2 + h ** 2) / 12.0, 0.5 * m * r ** 2])


def _translate(inertia, m, xyz):  # inertia.py:23-37 (translate_I), incl. its I[0][1] reuse for the xz term
    x, y, z = xyz
    out = np.zeros((3, 3))
    out[0, 0] = inertia[0, 0] + m * (y ** 2 + z ** 2)
    out[1, 1] = inertia[1, 1] + m * (x ** 2 + z ** 2)
    out[2, 2] = inertia[2, 2] + m * (x ** 2 + y ** 2)
    out[0, 1] = out[1, 0] = inertia[0, 1] + m * x * y
    out[0, 2] = out[2, 0] = inertia[0, 1] + m * x * z
    out[1, 2] = out[2, 1] = inertia[1, 2] + m * y * z
    return out


def crazyflie(thrust_noise_ratio=0.05, dt=1.0 / 200.0):
    """Returns the dict of airframe/motor constants consumed by `config.make_config`."""
    body = dict(l=0.03, w=0.03, h=0.004, m=0.005)
    payload = dict(l=0.035, w=0.02, h=0.008, m=0.01)
    arms = dict(l=0.022, w=0.005, h=0.005, m=0.001)
    motors = dict(h=0.02, r=0.0035, m=0.0015)
    props = dict(h=0.002, r=0.022, m=0.00075)
    motor_xyz = np.array([0.065 / 2, 0.065 / 2, 0.0])
    arm_angle = 45.0 / 180.0 * math.pi
    delta_y = motor_xyz[1] - body["w"] / 2.0
    arm_xyz = np.array([motor_xyz[0] - delta_y / (2 * math.tan(arm_angle)), motor_xyz[1] - delta_y / 2, 0.0])
    sign = np.array([[1, -1, -1, 1], [-1, -1, 1, 1], [1.0, 1.0, 1.0, 1.0]])  # FR, BR, BL, FL
    motors_coord = sign * motor_xyz[:, None]
    props_coord = motors_coord.copy()
    props_coord[2, :] += motors["h"] / 2.0 + props["h"]
    arms_coord = sign * arm_xyz[:, None]
    arm_angles = [-arm_angle, arm_angle, -arm_angle, arm_angle]

    links = [(_box_inertia(**body), body["m"], np.zeros(3), np.eye(3)),
             (_box_inertia(**payload), payload["m"], np.array([0.0, 0.0, (body["h"] + payload["h"]) / 2]), np.eye(3))]
    for i in range(4):
        a = arm_angles[i]
        rot = np.array([[math.cos(a), -math.sin(a), 0.0], [math.sin(a), math.cos(a), 0.0], [0.0, 0.0, 1.0]])
        links.append((_box_inertia(**arms), arms["m"], arms_coord[:, i].copy(), rot))
    for i in range(4):
        links.append((_cyl_inertia(**motors), motors["m"], motors_coord[:, i].copy(), np.eye(3)))
    for i in range(4):
        links.append((_cyl_inertia(**props), props["m"], props_coord[:, i].copy(), np.eye(3)))

    mass = float(np.sum([m for _, m, _, _ in links]))
    com = sum(m * xyz for _, m, xyz, _ in links) / mass
    total = np.zeros((3, 3))
    for inertia, m, xyz, rot in links:
        total += _translate(rot @ inertia @ rot.T, m, xyz - com)
    prop_pos = np.array([motors_coord[:, i] - com for i in range(4)])

    thrust_to_weight, torque_to_thrust = 1.9, 0.006
    thrust_max = GRAV * mass * thrust_to_weight * np.ones(4) / 4.0
    return dict(
        mass=mass, inertia=np.diagonal(total).copy(), arm=float(np.linalg.norm(motor_xyz[:2])),
        prop_pos=prop_pos, prop_cross=np.cross(prop_pos, [0.0, 0.0, 1.0]), prop_ccw=np.array([-1.0, 1.0, -1.0, 1.0]),
        thrust_max=thrust_max, torque_max=torque_to_thrust * thrust_max,
        motor_tau_up=4 * dt / (0.15 + 1e-6), motor_tau_down=4 * dt / (0.15 + 1e-6),
        motor_linearity=1.0, vel_damp=0.0, damp_omega_quadratic=0.0, omega_max=40.0, gravity=GRAV,
        thrust_noise_sigma=0.2 * thrust_noise_ratio, ou_theta=0.15, vxyz_max=3.0,
    )
