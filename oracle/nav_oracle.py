"""CPU/GPU-agnostic restatement of the two nav/ callers that consume the hot path's queries:

  planner_costs      Planner.calc_everything + body_to_world + get_state_cost + total_cost   nav/quad_plot.py:120-250
                     (with next_rotation nav/quad_helpers.py:186-199 and rot_matrix_to_vec nav/math_utils.py:115-156)
  measurement_loss   Estimator.measurement_fn                                                nav/estimator_helpers.py:293-327
                     (with vec_to_rot_matrix / skew_matrix / rot_x / nerf_matrix_to_ngp_torch / mahalanobis nav/math_utils.py:17-37,158-185)

TEST INFRASTRUCTURE ONLY: the product never imports this file.  The hot path enters as injected callables exactly as in the
reference (`density_fn`, `get_rays_fn`, `render_fn`: simulate.py:340-347), so the same restatement runs over the CPU oracle's field
(pinning the restatement against tests/golden/callers_nav.npz, which the reference's own nav/ code produced) and over the HIP
product's NavQueries / NativeNavQueries on the GPU.  Every tensor is created on the device of its inputs (the reference relies on
`torch.set_default_tensor_type('torch.cuda.FloatTensor')`, simulate.py:161).
"""
import numpy as np
import torch


# ------------------------------------------------------------------------------------------------------------------------
# rotations (nav/math_utils.py, nav/quad_helpers.py)
# ------------------------------------------------------------------------------------------------------------------------
def _hat(v):
    """skew_matrix (nav/math_utils.py:175-185): [..., 3] -> [..., 3, 3]"""
    S = torch.zeros(*v.shape[:-1], 3, 3, dtype=v.dtype, device=v.device)
    S[..., 0, 1], S[..., 0, 2] = -v[..., 2], v[..., 1]
    S[..., 1, 0], S[..., 1, 2] = v[..., 2], -v[..., 0]
    S[..., 2, 0], S[..., 2, 1] = -v[..., 1], v[..., 0]
    return S


def rotvec_to_matrix(r):
    """vec_to_rot_matrix (nav/math_utils.py:158-173): Rodrigues with the reference's `1e-10 + angle` guard"""
    theta = torch.norm(r, dim=-1, keepdim=True)
    K = _hat(r / (1e-10 + theta))
    theta = theta[..., None]
    return torch.eye(3, dtype=r.dtype, device=r.device) + torch.sin(theta) * K + (1 - torch.cos(theta)) * K @ K


def matrix_to_rotvec(R):
    """rot_matrix_to_vec (nav/math_utils.py:115-156), including its linearised acos near |x| = 1 (:120-132) and the zero-angle overwrite (:150)"""
    tr = torch.diagonal(R, dim1=-2, dim2=-1).sum(-1)
    x = (tr - 1) / 2
    eps = 1e-7
    slope = np.arccos(1 - eps) / eps
    inner = abs(x) <= 1 - eps
    outer = ~inner
    ang = torch.empty_like(x)
    ang[inner] = torch.acos(x[inner])
    sgn = torch.sign(x[outer])
    ang[outer] = torch.acos(sgn * (1 - eps)) - slope * sgn * (abs(x[outer]) - 1 + eps)
    ang = ang[..., None]
    axis = 1 / (2 * torch.sin(ang + 1e-10)) * torch.stack([R[..., 2, 1] - R[..., 1, 2], R[..., 0, 2] - R[..., 2, 0], R[..., 1, 0] - R[..., 0, 1]], dim=-1)
    axis[ang[..., 0] == 0] = torch.zeros(3, dtype=R.dtype, device=R.device)
    return ang * axis


def step_rotation(R, omega, dt):
    """next_rotation (nav/quad_helpers.py:186-199): R exp(hat(omega dt)), identity for a zero rate"""
    phi = omega * dt
    theta = torch.norm(phi, p=2)
    E = torch.eye(3, dtype=R.dtype, device=R.device)
    if theta != 0:
        K = _hat(phi / theta)
        E = E + torch.sin(theta) * K + (1 - torch.cos(theta)) * torch.matmul(K, K)
    return R @ E


# ------------------------------------------------------------------------------------------------------------------------
# planner (nav/quad_plot.py)
# ------------------------------------------------------------------------------------------------------------------------
def reduced_state(state):
    """full_to_reduced_state (nav/quad_plot.py:55-62): 18-vector -> (x, y, z, heading)"""
    R = state[6:15].reshape(3, 3)
    ex = R @ torch.tensor([1.0, 0, 0], dtype=state.dtype, device=state.device)
    return torch.cat([state[:3], torch.atan2(ex[1], ex[0])[None]]).detach()


def planner_initial_states(start_state, end_state, steps):
    """Planner.__init__ (nav/quad_plot.py:33-39): the straight-line initial guess, steps - 2 rows of (x, y, z, heading)"""
    s = torch.linspace(0, 1, steps, dtype=start_state.dtype, device=start_state.device)[1:-1, None]
    return (1 - s) * reduced_state(start_state) + s * reduced_state(end_state)


def robot_body(extent, nbins, dtype=torch.float32, device=None):
    """Planner.__init__ (nav/quad_plot.py:43-48): the body's point cloud, [prod(nbins), 3]"""
    axes = [torch.linspace(float(extent[k][0]), float(extent[k][1]), int(nbins[k]), dtype=dtype, device=device) for k in range(3)]
    return torch.stack(torch.meshgrid(*axes, indexing="ij"), dim=-1).reshape(-1, 3)


def planner_kinematics(states, initial_accel, start_state, end_state, cfg):
    """calc_everything (nav/quad_plot.py:120-196): differential-flatness reconstruction of the full trajectory.
    Returns pos, vel, accel, rot [S,3,3], omega, angular_accel, actions [S,4]."""
    dt = cfg["T_final"] / cfg["steps"]
    kw = dict(dtype=states.dtype, device=states.device)
    gravity = torch.tensor([0., 0., -cfg["g"]], **kw)
    ez = torch.tensor([0, 0, 1.0], **kw)
    p0, v0, R0, w0 = start_state[None, 0:3], start_state[None, 3:6], start_state[6:15].reshape(1, 3, 3), start_state[None, 15:]
    p1, v1, R1, w1 = end_state[None, 0:3], end_state[None, 3:6], end_state[6:15].reshape(1, 3, 3), end_state[None, 15:]
    Rn = step_rotation(R0, w0, dt)
    a_first = R0 @ ez * initial_accel[0] + gravity                     # :137-138
    a_second = Rn @ ez * initial_accel[1] + gravity
    v_second = v0 + a_first * dt
    v_third = v_second + a_second * dt
    p_second = p0 + v0 * dt
    p_third = p_second + v_second * dt
    p_fourth = p_third + v_third * dt
    pos = torch.cat([p0, p_second, p_third, p_fourth, states[2:, :3], p1], dim=0)          # :148
    vel = torch.cat([(pos[1:] - pos[:-1]) / dt, v1], dim=0)
    acc = (vel[1:] - vel[:-1]) / dt - gravity
    acc = torch.cat([acc, acc[-1, None, :]], dim=0)
    thrust = torch.norm(acc, dim=-1, keepdim=True)
    zb = (acc / thrust)[2:-1, :]                                        # rotations 0, 1 and the last are constrained (:166)
    heading = states[:, 3]
    plane = torch.stack([torch.sin(heading), -torch.cos(heading), torch.zeros_like(heading)], dim=-1)
    xb = torch.cross(zb, plane, dim=-1)
    xb = xb / torch.norm(xb, dim=-1, keepdim=True)
    yb = torch.cross(zb, xb, dim=-1)
    rot = torch.cat([R0, Rn, torch.stack([xb, yb, zb], dim=-1), R1], dim=0)
    omega = torch.cat([matrix_to_rotvec(rot[1:] @ rot[:-1].swapdims(-1, -2)) / dt, w1], dim=0)
    alpha = (omega[1:] - omega[:-1]) / dt
    alpha = torch.cat([alpha, alpha[-1, None, :]], dim=0)
    torque = (cfg["I"].to(**kw) @ alpha[..., None])[..., 0]
    return pos, vel, acc, rot, omega, alpha, torch.cat([thrust * cfg["mass"], torque], dim=-1)


def planner_costs(states, initial_accel, start_state, end_state, cfg, body, density_fn, epoch=0):
    """get_state_cost + total_cost (nav/quad_plot.py:224-254).  `density_fn` is simulate.py:343's lambda ([S,B,3] -> [S,B]).
    Returns dict(per_state, collision, total, points [S,B,3])."""
    pos, vel, acc, rot, omega, alpha, actions = planner_kinematics(states, initial_accel, start_state, end_state, cfg)
    thrust = actions[:, 0]
    torque = torch.norm(actions[:, 1:], dim=-1)
    speed = torch.sum(vel ** 2 + 1e-5, dim=-1) ** 0.5                   # :234
    points = (rot @ body.T + pos[..., None]).swapdims(-1, -2)           # body_to_world :216-222
    density = density_fn(points) ** 2                                   # :237
    collision = torch.mean(density * speed[:, None], dim=-1)
    if epoch < cfg["fade_out_epoch"]:                                   # :243-247
        s = torch.linspace(0, 1, collision.shape[0], dtype=collision.dtype, device=collision.device)
        collision = collision * torch.sigmoid(cfg["fade_out_sharpness"] * (epoch / cfg["fade_out_epoch"] - s))
    per_state = 1000 * thrust ** 2 + 0.01 * torque ** 4 + collision * 1e6
    return dict(per_state=per_state, collision=collision * 1e6, total=torch.mean(per_state), points=points)


# ------------------------------------------------------------------------------------------------------------------------
# pose filter (nav/estimator_helpers.py)
# ------------------------------------------------------------------------------------------------------------------------
def camera_pose_from_state(state):
    """measurement_fn's pose chain (nav/estimator_helpers.py:301-309): R(state[6:9]) -> rot_x(pi/2) R -> the NGP axis convention
    (nerf_matrix_to_ngp_torch, nav/math_utils.py:25-37) -> 4x4"""
    kw = dict(dtype=torch.float32, device=state.device)
    phi = torch.tensor(np.pi / 2)
    about_x = torch.tensor([[1., 0., 0.], [0., torch.cos(phi), -torch.sin(phi)], [0., torch.sin(phi), torch.cos(phi)]], **kw)
    flip_sign = torch.tensor([[1, 0, 0], [0, -1, 0], [0, 0, -1]], **kw)
    cycle = torch.tensor([[0, 1, 0], [0, 0, 1], [1, 0, 0]], **kw)
    R = about_x @ rotvec_to_matrix(state[6:9])[:3, :3]
    pose = torch.eye(4, **kw)
    pose[:3, :3] = cycle @ R @ flip_sign
    pose[:3, 3] = cycle @ state[:3]
    return pose


def measurement_loss(state, start_state, sig, target, batch, get_rays_fn, render_fn):
    """Estimator.measurement_fn (nav/estimator_helpers.py:293-327): mse(render(rays(pose(state)))[batch], target[batch]) + Mahalanobis(state, start).
    target [H,W,3]; batch [n,2] integer (row, column) pairs; get_rays_fn(pose [1,4,4]) -> dict(rays_o, rays_d [1,H*W,3])."""
    delta = state - start_state
    process = delta @ torch.inverse(sig) @ delta                        # nav/math_utils.py:21-23
    H, W, _ = target.shape
    rays = get_rays_fn(camera_pose_from_state(state).reshape(1, 4, 4))
    rows, cols = batch[:, 0], batch[:, 1]
    o = rays["rays_o"].reshape(H, W, -1)[rows, cols]
    d = rays["rays_d"].reshape(H, W, -1)[rows, cols]
    rgb = render_fn(o.reshape(1, -1, 3), d.reshape(1, -1, 3))["image"].reshape(-1, 3)
    return torch.nn.functional.mse_loss(rgb, target[rows, cols]) + process


def measurement_hessian(state, start_state, sig, target, batch, get_rays_fn, render_fn):
    """the call of nav/estimator_helpers.py:384"""
    return torch.autograd.functional.hessian(lambda x: measurement_loss(x, start_state, sig, target, batch, get_rays_fn, render_fn), state)


def state18(pos, rotvec=(0., 0., 0.), dtype=torch.float32, device=None):
    """simulate.py:251-263: position + zero rates + rotation matrix (row-major) + zero rates"""
    rates = torch.zeros(3, dtype=dtype, device=device)
    R = rotvec_to_matrix(torch.tensor(rotvec, dtype=dtype, device=device))
    return torch.cat([torch.tensor(pos, dtype=dtype, device=device), rates, R.reshape(-1), rates], dim=0)
