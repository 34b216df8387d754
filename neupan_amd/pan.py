"""`PAN` -- host-side mirror of the reference's PAN module (neupan/blocks/pan.py:28-274) over
the HIP kernels of libneupan_amd.so.

Same constructor, same `forward(nom_s, nom_u, ref_s, ref_us, obs_points, point_velocities)`,
same attributes other reference code reads (`min_distance`, `dune_points`, `nrmp_points`,
`nrmp_layer.adjust_parameters`, `nrmp_layer.update_adjust_parameters_value`), so it can be
installed where `neupan.blocks.PAN` is imported (neupan/neupan.py:24, :84) -- see
INTEGRATION.md.  New: `forward_batch` plans B independent scenes in one call, which is where
the GPU earns its keep.

torch is used for device memory and streams only; all arithmetic of the path runs in the
HIP kernels (neupan_amd/csrc/*.hip) reached through the C ABI (include/neupan_amd.h).
There is no CPU fallback: constructing a PAN without the built library or without a GPU raises.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
from math import inf
from typing import Optional

import numpy as np
import torch

from . import _lib
from ._lib import KIN, NeupanAmdError, NpaConfig, NpaDuneWeights, check

_LINEAR = (0, 3, 5, 8, 10, 13)      # Linear layers in ObsPointNet.MLP (obs_point_net.py:31-46)
_NORM = (1, 6, 11)                  # LayerNorm layers


def _resolve_checkpoint(path):
    """Search order of the reference's util.file_check (util/__init__.py:58-94): as given,
    relative to the script directory, relative to the cwd.  Raises instead of prompting."""
    if path is None or path == "None":
        raise FileNotFoundError("dune_checkpoint is required (this build never trains or prompts; "
                                "reference behaviour at neupan/blocks/dune.py:184-207 is to block on input())")
    cands = [path, os.path.join(sys.path[0], path), os.path.join(os.getcwd(), path)]
    for c in cands:
        if os.path.isfile(c):
            return os.path.abspath(c)
    raise FileNotFoundError(f"DUNE checkpoint not found: {path}")


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


class _NrmpFacade:
    """The slice of `NRMP` other reference code touches (neupan/neupan.py:344-359, 377-379):
    adjust_parameters + update_adjust_parameters_value (nrmp.py:170-217)."""

    def __init__(self, pan, q_s, p_u, eta, d_max, d_min, no_obs):
        self._pan = pan
        self.no_obs = no_obs
        if isinstance(q_s, (list, tuple, np.ndarray)):
            q = np.array(q_s, dtype=np.float32).flatten()
            if q.shape[0] != 3:
                raise ValueError(f"q_s must be a scalar or a 3-element list/array, got {q.shape[0]} elements")
            self.q_s = torch.tensor(q.reshape(3, 1), dtype=torch.float32)
        else:
            self.q_s = torch.tensor(float(q_s), dtype=torch.float32)
        self.p_u = torch.tensor(float(p_u), dtype=torch.float32)
        self.eta = torch.tensor(float(eta), dtype=torch.float32)
        self.d_max = torch.tensor(float(d_max), dtype=torch.float32)
        self.d_min = torch.tensor(float(d_min), dtype=torch.float32)
        self.obstacle_points = None

    @property
    def adjust_parameters(self):
        return [self.q_s, self.p_u] if self.no_obs else [self.q_s, self.p_u, self.eta, self.d_max, self.d_min]

    def q_s3(self):
        q = self.q_s.detach().reshape(-1).tolist()
        return q * 3 if len(q) == 1 else q

    def update_adjust_parameters_value(self, **kwargs):
        if "q_s" in kwargs:
            v = kwargs["q_s"]
            if self.q_s.dim() == 0:
                if isinstance(v, (list, tuple, np.ndarray)):
                    v = v[0]          # reference prints and uses the first element (nrmp.py:191-193)
                self.q_s = torch.tensor(float(v), dtype=torch.float32)
            else:
                if not isinstance(v, (list, tuple, np.ndarray)) or len(np.array(v).flatten()) != 3:
                    raise ValueError("q_s must be a 3-element list/array for a vector-initialised planner")
                self.q_s = torch.tensor(np.array(v, dtype=np.float32).reshape(3, 1))
        for k in ("p_u", "eta", "d_max", "d_min"):
            if k in kwargs:
                setattr(self, k, torch.tensor(float(kwargs[k]), dtype=torch.float32))
        self._pan._push_adjust()

    @property
    def points(self):
        return self.obstacle_points


class _DuneFacade:
    def __init__(self, checkpoint, pan=None):
        self.abs_checkpoint_path = checkpoint
        self.obstacle_points = None
        self.min_distance = inf
        self._pan = pan
        self.full_model_name = None

    def train_dune(self, train_kwargs=None):
        """dune.py:174-182: train an ObsPointNet for this robot's polygon and return the checkpoint path
        (labels by the HIP labeller instead of ECOS, neupan_amd/dune_train.py).  Construct a new PAN with
        `dune_checkpoint=<that path>` to plan with it (the reference asks interactively, dune.py:160-166)."""
        from .dune_train import DuneTrain
        kw = dict(train_kwargs or {})
        pan = self._pan
        name = kw.pop("model_name", getattr(pan.robot, "name", None) or "robot")
        kw.pop("direct_train", None)
        # dune_train.py:66-69 writes to <script dir>/model/<name>; `save_dir` (ours) overrides the root
        path = os.path.join(kw.pop("save_dir", None) or sys.path[0] or os.getcwd(), "model", name)
        tr = DuneTrain(None, np.asarray(pan.robot.G, dtype=np.float32), np.asarray(pan.robot.h, dtype=np.float32), path,
                       device=pan.device)
        self.full_model_name = tr.start(**kw)
        return self.full_model_name

    @property
    def points(self):
        return self.obstacle_points


class PAN(torch.nn.Module):
    """Drop-in for neupan.blocks.PAN (constructor: pan.py:43-56)."""

    def __init__(self, receding=10, step_time=0.1, robot=None, iter_num=2, dune_max_num=100, nrmp_max_num=10,
                 dune_checkpoint=None, iter_threshold=0.1, adjust_kwargs=None, train_kwargs=None, device=None,
                 **kwargs):
        super().__init__()
        if robot is None:
            raise ValueError("robot parameter is required and cannot be None")      # dune.py:37-38
        adjust_kwargs = dict(adjust_kwargs or {})
        self.robot, self.T, self.dt = robot, int(receding), float(step_time)
        self.iter_num, self.iter_threshold = int(iter_num), float(iter_threshold)
        self.nrmp_max_num, self.dune_max_num = int(nrmp_max_num), int(dune_max_num)
        self.no_obs = (self.nrmp_max_num == 0 or self.dune_max_num == 0)
        self.ro_obs = float(adjust_kwargs.get("ro_obs", 400))
        self.bk = float(adjust_kwargs.get("bk", 0.1))
        self.nrmp_layer = _NrmpFacade(self, adjust_kwargs.get("q_s", 1.0), adjust_kwargs.get("p_u", 1.0),
                                      adjust_kwargs.get("eta", 10.0), adjust_kwargs.get("d_max", 1.0),
                                      adjust_kwargs.get("d_min", 0.1), self.nrmp_max_num == 0)

        if not torch.cuda.is_available():
            raise NeupanAmdError("neupan_amd.PAN needs a ROCm GPU (torch.cuda.is_available() is False); "
                                 "there is no CPU fallback")
        self.device = torch.device(device if device is not None else "cuda")
        self._lib = _lib.load()

        G = np.asarray(robot.G, dtype=np.float32)
        h = np.asarray(robot.h, dtype=np.float32).reshape(-1)
        self.E = int(G.shape[0])
        if robot.kinematics not in KIN:
            raise ValueError("kinematics currently only supports acker, diff or omni")
        cfg = NpaConfig()
        cfg.receding, cfg.iter_num = self.T, self.iter_num
        cfg.dune_max_num, cfg.nrmp_max_num, cfg.edge_num = self.dune_max_num, self.nrmp_max_num, self.E
        cfg.kinematics = KIN[robot.kinematics]
        cfg.iter_threshold = self.iter_threshold
        cfg.step_time = self.dt
        cfg.wheelbase = float(robot.L) if getattr(robot, "L", None) is not None else 0.0
        sb = np.asarray(robot.speed_bound, dtype=np.float64).reshape(-1)
        ab = np.asarray(robot.acce_bound, dtype=np.float64).reshape(-1)
        for k in range(2):
            cfg.speed_bound[k], cfg.acce_bound[k] = sb[k], ab[k]
        cfg.ro_obs, cfg.bk = self.ro_obs, self.bk
        if self.E > _lib.NPA_MAX_E:
            raise NeupanAmdError(f"robot polygon has {self.E} edges; this build supports <= {_lib.NPA_MAX_E}")
        for e in range(self.E):
            cfg.G[e][0], cfg.G[e][1], cfg.h[e] = float(G[e, 0]), float(G[e, 1]), float(h[e])
        self._cfg = cfg
        self._fill_adjust(cfg)

        wts = None
        self._untrained = False
        train_kwargs = dict(train_kwargs or {})
        if not self.no_obs and dune_checkpoint in (None, "None") and train_kwargs.get("direct_train", False):
            # the reference's training workflow (example/dune_train/*.yaml: no checkpoint, train.direct_train: true;
            # dune.py:152-156 constructs the planner and then trains): no obstacle stage exists until train_dune() has
            # produced a checkpoint and a planner is constructed with it -- forward raises until then
            self._untrained = True
            self.dune_layer = _DuneFacade(None, self)
            self._h = C.c_void_p()
            self._B = 0
            self._ws = self._state = self._last = None
            self.printed = False
            return
        if not self.no_obs:
            path = _resolve_checkpoint(dune_checkpoint)
            sd = torch.load(path, map_location="cpu")                    # dune.py:142
            keep = []
            wts = NpaDuneWeights()

            def arr(key, shape):
                a = np.ascontiguousarray(sd[key].detach().to(torch.float32).numpy())
                if tuple(a.shape) != shape:
                    raise ValueError(f"checkpoint tensor {key} has shape {a.shape}, expected {shape}")
                keep.append(a)
                return a.ctypes.data

            shapes = [(32, 2)] + [(32, 32)] * 4 + [(self.E, 32)]
            for i, li in enumerate(_LINEAR):
                wts.lin_w[i] = arr(f"MLP.{li}.weight", shapes[i])
                wts.lin_b[i] = arr(f"MLP.{li}.bias", (shapes[i][0],))
            for i, li in enumerate(_NORM):
                wts.ln_w[i] = arr(f"MLP.{li}.weight", (32,))
                wts.ln_b[i] = arr(f"MLP.{li}.bias", (32,))
            self._wkeep = keep
            self.dune_layer = _DuneFacade(path, self)
        else:
            self.dune_layer = None

        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            check(self._lib.npa_create(C.byref(cfg), C.byref(wts) if wts is not None else None, C.byref(self._h)),
                  "npa_create")
        self._B = 0
        self._ws = self._state = None
        self._last = None
        self.printed = False

    # ------------------------------------------------------------------ plumbing
    def _fill_adjust(self, cfg):
        q = self.nrmp_layer.q_s3()
        for k in range(3):
            cfg.q_s[k] = q[k]
        cfg.p_u = float(self.nrmp_layer.p_u)
        cfg.eta, cfg.d_max, cfg.d_min = float(self.nrmp_layer.eta), float(self.nrmp_layer.d_max), float(self.nrmp_layer.d_min)

    def _push_adjust(self):
        self._fill_adjust(self._cfg)
        q = (C.c_float * 3)(*self.nrmp_layer.q_s3())
        check(self._lib.npa_set_adjust(self._h, C.byref(q), self._cfg.p_u, self._cfg.eta, self._cfg.d_max,
                                       self._cfg.d_min), "npa_set_adjust")

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None and self._h.value:
                self._lib.npa_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    def _get_buffers(self, B):
        if B != self._B:
            wsb = self._lib.npa_workspace_bytes(self._h, B)
            stb = self._lib.npa_state_bytes(self._h, B)
            self._ws = torch.empty(wsb, dtype=torch.uint8, device=self.device)
            self._state = torch.zeros(stb, dtype=torch.uint8, device=self.device)
            self._B = B
        return self._ws, self._state

    def reset_stop_state(self):
        """Forget the previous iterate (the reference keeps it for the life of the PAN object)."""
        if self._state is not None:
            self._state.zero_()

    def _dev(self, t, shape=None):
        if t is None:
            return None
        if not isinstance(t, torch.Tensor):
            t = torch.as_tensor(np.asarray(t))
        t = t.detach().to(device=self.device, dtype=torch.float32).contiguous()
        if shape is not None and tuple(t.shape) != tuple(shape):
            raise ValueError(f"expected tensor of shape {tuple(shape)}, got {tuple(t.shape)}")
        return t

    # ------------------------------------------------------------------ batched entry
    def forward_begin(self, nom_s, nom_u, ref_s, ref_us, points=None, velocities=None, n_points=None, reset_state=False,
                      out_u=None):
        """Stage one batch (see forward_batch for shapes) and start a forward on the current stream: follow with
        forward_iter(k) for k in range(iter_num) and forward_end()."""
        if self._untrained:
            raise NeupanAmdError("this planner was constructed without a DUNE checkpoint (train.direct_train): call "
                                 "dune_layer.train_dune(...) and construct a planner with dune_checkpoint=<the file it returns>")
        T, M = self.T, self.nrmp_max_num
        nom_s = self._dev(nom_s)
        B = nom_s.shape[0]
        nom_s = self._dev(nom_s, (B, 3, T + 1))
        nom_u, ref_s, ref_us = self._dev(nom_u, (B, 2, T)), self._dev(ref_s, (B, 3, T + 1)), self._dev(ref_us, (B, T))
        use_pts = points is not None and not self.no_obs
        n_stride = 0
        if use_pts:
            points = self._dev(points)
            if points.dim() != 3 or points.shape[0] != B or points.shape[1] != 2:
                raise ValueError(f"points must have shape (B,2,N), got {tuple(points.shape)}")
            n_stride = points.shape[2]
            if n_stride == 0:
                use_pts = False
        if use_pts:
            if velocities is not None:
                velocities = self._dev(velocities, (B, 2, n_stride))
            if n_points is not None:
                n_points = torch.as_tensor(n_points).to(device=self.device, dtype=torch.int32).contiguous()
            if n_stride > self.dune_max_num and not self.printed:
                print(f"down sample the obs points from {n_stride} to {self.dune_max_num}")   # pan.py:172
                self.printed = True
        else:
            points = velocities = n_points = None
        ws, state = self._get_buffers(B)
        dev = self.device
        out_s = torch.empty((B, 3, T + 1), dtype=torch.float32, device=dev)
        if out_u is None:
            out_u = torch.empty((B, 2, T), dtype=torch.float32, device=dev)
        elif tuple(out_u.shape) != (B, 2, T) or out_u.dtype != torch.float32 or not out_u.is_contiguous() or out_u.device != dev:
            raise ValueError("out_u must be a contiguous float32 tensor of shape (B, 2, T) on the planner's device")
        out_d = torch.empty((B, 1, max(T, 1)), dtype=torch.float32, device=dev) if M > 0 and self.dune_max_num > 0 else None
        out_md = torch.empty((B,), dtype=torch.float32, device=dev)
        # every scene's QP writes these in PAN iteration 0 at the latest: no fill kernels needed
        out_it = torch.empty((B,), dtype=torch.int32, device=dev)
        out_np = torch.empty((B, 2, M), dtype=torch.float32, device=dev) if not self.no_obs else None
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            check(self._lib.npa_forward_begin(
                self._h, B, max(n_stride, 1), _ptr(nom_s), _ptr(nom_u), _ptr(ref_s), _ptr(ref_us), _ptr(points),
                _ptr(velocities), _ptr(n_points), _ptr(out_s), _ptr(out_u), _ptr(out_d), _ptr(out_md), _ptr(out_it),
                _ptr(out_np), _ptr(ws), ws.numel(), _ptr(state), state.numel(), C.c_void_p(stream),
                2 if reset_state else 0), "npa_forward_begin")
        # keep inputs alive until the stream has consumed them.  (DUNE.min_distance keeps its last value over calls without
        # points -- dune.py:97-98 runs only with points, pan.py:246-252 reads the attribute: the kernel carries that value
        # in the scene's state record, so out_md already is the persistent value, read or not in between.)
        self._last = dict(points=points, velocities=velocities, n_points=n_points, min_distance=out_md,
                          nrmp_points=out_np, used_points=use_pts, hold=(nom_s, nom_u, ref_s, ref_us))
        self._pending = dict(opt_s=out_s, opt_u=out_u, opt_d=out_d, min_distance=out_md, iters=out_it, nrmp_points=out_np)

    def _workspace_views(self):
        """Views into the workspace of the batch in progress: dict(cur_s, cur_u, mu, lam, pts, dist, count) (stream-ordered)."""
        B, T, M, E = self._B, self.T, max(self.nrmp_max_num, 1), self.E
        off = (C.c_size_t * 8)()
        check(self._lib.npa_workspace_layout(self._h, B, off, 8), "npa_workspace_layout")
        ws = self._ws

        def view(i, n, dtype, shape):
            return ws[off[i]:off[i] + 4 * n].view(dtype).reshape(shape)
        return dict(cur_s=view(0, B * 3 * (T + 1), torch.float32, (B, 3, T + 1)), cur_u=view(1, B * 2 * T, torch.float32, (B, 2, T)),
                    mu=view(3, B * (T + 1) * M * E, torch.float32, (B, T + 1, M, E)),
                    lam=view(4, B * (T + 1) * M * 2, torch.float32, (B, T + 1, M, 2)),
                    pts=view(5, B * (T + 1) * M * 2, torch.float32, (B, T + 1, M, 2)),
                    dist=view(6, B * (T + 1) * M, torch.float32, (B, T + 1, M)),
                    count=view(7, B * (T + 1), torch.int32, (B, T + 1)))

    def forward_iter(self, k):
        """Enqueue PAN iteration k (selection + QP launches) of the forward started by forward_begin."""
        with torch.cuda.device(self.device):       # the launches must see the device of the handle
            check(self._lib.npa_forward_iter(self._h, int(k)), "npa_forward_iter")

    def forward_end(self):
        """Close the forward; returns the output dict (device tensors, valid in stream order)."""
        with torch.cuda.device(self.device):
            check(self._lib.npa_forward_end(self._h), "npa_forward_end")
        out, self._pending = self._pending, None
        self.last_out = out
        return out

    def forward_batch(self, nom_s, nom_u, ref_s, ref_us, points=None, velocities=None, n_points=None, reset_state=False):
        """Plan B independent scenes (reset_state: forget the stop criterion's previous iterate first, inside the staging launch).  Shapes: nom_s (B,3,T+1) nom_u (B,2,T) ref_s (B,3,T+1)
        ref_us (B,T) points (B,2,N)|None velocities (B,2,N)|None n_points (B,) int32|None.
        Returns dict(opt_s, opt_u, opt_d|None, min_distance (B,), iters (B,), nrmp_points (B,2,M)|None)."""
        self.forward_begin(nom_s, nom_u, ref_s, ref_us, points, velocities, n_points, reset_state=reset_state)
        # (the K iterations and the end under ONE device context, library calls direct: a serving loop issues this per
        # batch in flight, and the per-iteration Python was half of a step's host time)
        lib, h = self._lib, self._h
        with torch.cuda.device(self.device):
            try:
                # (self.iter_num, not the handle's creation-time K: the reference's PAN.iter_num is an attribute callers may lower)
                for k in range(self.iter_num):
                    rc = lib.npa_forward_iter(h, k)
                    if rc:
                        check(rc, "npa_forward_iter")
            except BaseException:
                # close the call on the handle, or every later forward_begin fails with "previous forward not ended"
                lib.npa_forward_end(h)
                self._pending = None
                raise
            check(lib.npa_forward_end(h), "npa_forward_end")
        out, self._pending = self._pending, None
        self.last_out = out
        self._calls = getattr(self, "_calls", 0) + 1
        if (self._calls & 63) == 0:                  # (a host read of one pinned word: no synchronisation)
            self.check_audit()
        return out

    def make_step(self, nom_s, nom_u, ref_s, ref_us, points=None, velocities=None, n_points=None, reset_state=False,
                  graph=False, out_u=None, reset_every_step=False):
        """Serving-loop form of forward_batch: validate and convert the arguments ONCE and return `step()`, which plans
        the batch with ONE library call (npa_forward_batch_flags) on the current stream and returns the same dict of
        output tensors every time -- they are allocated here and REUSED (forward_batch returns fresh tensors per call,
        like the reference; a loop that owns its buffers does not need that, and the per-call conversions, allocations and
        a dozen ctypes calls were a third of a step's wall time).  The input tensors are read in place at every step():
        refresh them with copy_() between steps.  out_u: a caller-owned (B, 2, T) tensor the controls are written to (e.g. a
        slot of a gather staging buffer, neupan_amd.serve.ControlGatherer).
        reset_state applies to the priming forward run here ONLY (a fresh planner); step() then carries the stop
        criterion's memory, the QP warm start and the persisted min_distance from call to call like the reference's PAN
        object does.  reset_every_step=True makes EVERY step() start from a cleared state (a benchmark whose steps must
        all do the same work; bench.py).
        step() re-validates what it captured: the input tensors must still live at the addresses they had here (a
        tensor that was resize_()d / set_() / moved raises instead of planning stale memory; REBINDING a Python name to a
        new tensor cannot be seen -- refresh inputs with copy_()), and every 64th call polls the margin audit
        (check_audit) and warns once if the geometric-key margin was ever violated.
        graph=True: the step's launches (staging + K x {selection, QP}) are recorded once into a HIP graph and step()
        replays it: one graph launch instead of 1 + 2K kernel launches on the host (the kernels, their order and their
        results are the same; the per-launch profiling events of profile() do not exist inside a graph)."""
        # the caller's own device tensors (read in place by every step): their address and shape are re-checked per call
        watched = [(t, t.data_ptr(), tuple(t.shape)) for t in (nom_s, nom_u, ref_s, ref_us, points, velocities, n_points, out_u)
                   if isinstance(t, torch.Tensor) and t.device == self.device and t.is_contiguous()
                   and t.dtype in (torch.float32, torch.int32)]
        self.forward_begin(nom_s, nom_u, ref_s, ref_us, points, velocities, n_points,
                           reset_state=(reset_state or reset_every_step), out_u=out_u)
        for k in range(self.iter_num):               # (the first step also runs here: buffers and the handle are warm)
            self.forward_iter(k)
        out = self.forward_end()
        last = self._last
        nom_s, nom_u, ref_s, ref_us = last["hold"]
        pts, vel, npt = last["points"], last["velocities"], last["n_points"]
        B = nom_s.shape[0]
        ws, state = self._get_buffers(B)
        n_stride = max(pts.shape[2], 1) if pts is not None else 1
        lib, h, dev = self._lib, self._h, self.device
        args = (h, B, n_stride, _ptr(nom_s), _ptr(nom_u), _ptr(ref_s), _ptr(ref_us), _ptr(pts), _ptr(vel), _ptr(npt),
                _ptr(out["opt_s"]), _ptr(out["opt_u"]), _ptr(out["opt_d"]), _ptr(out["min_distance"]), _ptr(out["iters"]),
                _ptr(out["nrmp_points"]), _ptr(ws), ws.numel(), _ptr(state), state.numel())
        flags = 2 if reset_every_step else 0
        fn = lib.npa_forward_batch_flags
        # what step() re-validates per call: (tensor, address captured here) of every buffer the library call reads or writes
        held = [t for t in (nom_s, nom_u, ref_s, ref_us, pts, vel, npt, out["opt_s"], out["opt_u"], out["opt_d"],
                            out["min_distance"], out["iters"], out["nrmp_points"], ws, state) if t is not None]
        held_ptrs = tuple(t.data_ptr() for t in held)
        calls = [0]

        def validate():
            if tuple(t.data_ptr() for t in held) != held_ptrs or \
                    any(t.data_ptr() != p0 or tuple(t.shape) != sh for t, p0, sh in watched):
                raise NeupanAmdError("PAN.make_step: a tensor captured at prepare time has moved or changed shape (resize_ / "
                                     "set_ / a new workspace): step() would plan stale memory -- make the step again")
            calls[0] += 1
            if (calls[0] & 63) == 0:
                self.check_audit()
        cur_stream = torch.cuda.current_stream
        idx = dev.index if dev.index is not None else torch.cuda.current_device()

        cur_dev = torch.cuda.current_device
        if graph:
            g = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(cur_stream(dev))
            with torch.cuda.device(dev):
                with torch.cuda.graph(g, stream=side):
                    rc = fn(*args, C.c_void_p(cur_stream(dev).cuda_stream), flags)
            check(rc, "npa_forward_batch_flags (graph capture)")
            cur_stream(dev).wait_stream(side)

            def step_graph():
                validate()
                g.replay()
                self._last = last
                self.last_out = out
                return out
            step_graph.device_index = idx
            step_graph.graph = g
            return step_graph

        def step():
            validate()
            if step.pre_issue is not None:           # (e.g. the upload of this step's inputs, on the step's stream)
                step.pre_issue()
            if cur_dev() != idx:                     # the launches must see the device of the handle
                with torch.cuda.device(dev):
                    rc = fn(*args, C.c_void_p(cur_stream(dev).cuda_stream), flags)
            else:
                rc = fn(*args, C.c_void_p(cur_stream(dev).cuda_stream), flags)
            if rc:
                check(rc, "npa_forward_batch_flags")
            self._last = last
            self.last_out = out
            return out

        def finish():
            self._last = last
            self.last_out = out
            return out
        step.device_index = idx
        step.pre_issue = None                        # optional callable run on the step's stream in front of every issue
        # what StepGroup needs to issue this step inside a breadth-first burst (npa_forward_batch_group)
        step.group_call = dict(args=args, flags=flags, iter_num=self.iter_num, validate=validate, finish=finish, device=dev, lib=lib)
        return step

    def forward_batch_trace(self, nom_s, nom_u, ref_s, ref_us, points=None, velocities=None, n_points=None):
        """forward_batch that also returns the controls after every PAN iteration: out["trace_u"] (B, K, 2, T), the
        working nominal copied out of the workspace behind each iteration's QP (parity tooling: how a deviation from
        the oracle grows with the iteration count)."""
        self.forward_begin(nom_s, nom_u, ref_s, ref_us, points, velocities, n_points)
        B, T = self._B, self.T
        wsf = self._ws.view(torch.float32)
        off_u = (B * 3 * (T + 1) + 3) // 4 * 4                  # cur_u follows cur_s (pan_common.h: npa_scratch_layout)
        us, ss, ps, qi, ms, ls = [], [], [], [], [], []
        views = self._workspace_views() if not self.no_obs else None
        qoff = self._lib.npa_workspace_qp_info_offset(self._h, B)
        qview = self._ws[qoff:qoff + B * 16 * 8].view(torch.float64).reshape(B, 16)
        for k in range(self.iter_num):
            self.forward_iter(k)
            us.append(wsf[off_u:off_u + B * 2 * T].clone().reshape(B, 2, T))
            ss.append(wsf[:B * 3 * (T + 1)].clone().reshape(B, 3, T + 1))
            qi.append(qview.clone())
            if views is not None:
                ps.append(views["pts"].clone()); ms.append(views["mu"].clone()); ls.append(views["lam"].clone())
        out = self.forward_end()
        out["trace_qp_info"] = torch.stack(qi, dim=1)    # (B, K, 16): solver diagnostics of every iteration's QP (include/neupan_amd.h)
        out["trace_u"] = torch.stack(us, dim=1)
        out["trace_s"] = torch.stack(ss, dim=1)          # (B, K, 3, T+1): the states each iteration hands to the next
        # (B, K, T+1, M, 2): the points of the rows the selection emitted in every iteration (slice 0 is only redone in the
        # first iteration and keeps its rows afterwards)
        out["trace_pts"] = torch.stack(ps, dim=1) if ps else None
        # (B, K, T+1, M, E) / (B, K, T+1, M, 2): the mu / lam rows of the same slices -- with trace_pts everything the QP of that
        # iteration was built from (parity tooling: the oracle's solver on the kernel's rows)
        out["trace_mu"] = torch.stack(ms, dim=1) if ms else None
        out["trace_lam"] = torch.stack(ls, dim=1) if ls else None
        return out

    # ------------------------------------------------------------------ reference signature
    def forward(self, nom_s, nom_u, ref_s, ref_us, obs_points=None, point_velocities=None):
        """pan.py:109-147 for one scene: (3,T+1),(2,T),(3,T+1),(T,),(2,N)|None,(2,N)|None ->
        (opt_s (3,T+1), opt_u (2,T), opt_d (1,T)|None)."""
        un = lambda t: None if t is None else (t if isinstance(t, torch.Tensor) else torch.as_tensor(np.asarray(t))).unsqueeze(0)
        if any(p.requires_grad for p in self.nrmp_layer.adjust_parameters):
            # LON use (example/LON/LON_corridor.py): outputs stay connected to the adjust parameters
            s, u, d = self.forward_batch_grad(un(nom_s), un(nom_u), un(ref_s), un(ref_us), un(obs_points), un(point_velocities))
            return s[0], u[0], (None if self.no_obs else d[0])
        out = self.forward_batch(un(nom_s), un(nom_u), un(ref_s), un(ref_us), un(obs_points), un(point_velocities))
        d = None if self.no_obs or out["opt_d"] is None else out["opt_d"][0]
        return out["opt_s"][0], out["opt_u"][0], d

    # ------------------------------------------------------------------ attributes of the reference class
    def current_min_distance(self):
        """(B,) tensor: DUNE.min_distance per scene with the reference's persistence -- a scene without points in the
        last call keeps the value of its last call WITH points (dune.py:97-98), +inf if there never was one."""
        if self.no_obs or self._last is None:
            return None
        return self._last["min_distance"]

    @property
    def min_distance(self):
        md = self.current_min_distance()
        if md is None:
            return inf
        return md[0] if md.shape[0] == 1 else md

    @property
    def dune_points(self):
        """Points DUNE considered at t=0 (after decimation), numpy (2,N) -- pan.py:255-261."""
        if self.no_obs or self._last is None or not self._last["used_points"]:
            return None
        p = self._last["points"][0]
        n = p.shape[1] if self._last["n_points"] is None else int(self._last["n_points"][0])
        p = p[:, :n].cpu().numpy()
        if n > self.dune_max_num:
            p = p[:, np.linspace(0, n - 1, self.dune_max_num).astype(int)]
        return p

    @property
    def nrmp_points(self):
        """Points NRMP considered (first M sorted points of slice 0), numpy -- pan.py:264-269."""
        if self.no_obs or self._last is None or not self._last["used_points"]:
            return None
        p = self._last["points"][0]
        n = p.shape[1] if self._last["n_points"] is None else int(self._last["n_points"][0])
        k = min(n, self.dune_max_num, self.nrmp_max_num)
        return self._last["nrmp_points"][0, :, :k].cpu().numpy()

    # ------------------------------------------------------------------ stage access (tests, profiling)
    def dune_stage(self, nom_s, points, velocities=None, n_points=None):
        T, M, E = self.T, self.nrmp_max_num, self.E
        nom_s, points = self._dev(nom_s), self._dev(points)
        B, N = nom_s.shape[0], points.shape[2]
        velocities = self._dev(velocities, (B, 2, N)) if velocities is not None else None
        if n_points is not None:
            n_points = torch.as_tensor(n_points).to(device=self.device, dtype=torch.int32).contiguous()
        dev = self.device
        mu = torch.zeros((B, T + 1, M, E), dtype=torch.float32, device=dev)
        lam = torch.zeros((B, T + 1, M, 2), dtype=torch.float32, device=dev)
        pts = torch.zeros((B, T + 1, M, 2), dtype=torch.float32, device=dev)
        dist = torch.zeros((B, T + 1, M), dtype=torch.float32, device=dev)
        cnt = torch.zeros((B, T + 1), dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            check(self._lib.npa_dune_stage(self._h, B, N, _ptr(nom_s), _ptr(points), _ptr(velocities), _ptr(n_points),
                                           _ptr(mu), _ptr(lam), _ptr(pts), _ptr(dist), _ptr(cnt), C.c_void_p(stream)),
                  "npa_dune_stage")
        torch.cuda.synchronize(dev)
        return dict(mu=mu, lam=lam, pts=pts, dist=dist, count=cnt)

    def nrmp_stage(self, nom_s, nom_u, ref_s, ref_us, stage=None):
        T, M = self.T, self.nrmp_max_num
        nom_s = self._dev(nom_s)
        B = nom_s.shape[0]
        nom_u, ref_s, ref_us = self._dev(nom_u, (B, 2, T)), self._dev(ref_s, (B, 3, T + 1)), self._dev(ref_us, (B, T))
        dev = self.device
        out_s = torch.empty((B, 3, T + 1), dtype=torch.float32, device=dev)
        out_u = torch.empty((B, 2, T), dtype=torch.float32, device=dev)
        out_d = torch.empty((B, 1, T), dtype=torch.float32, device=dev)
        info = torch.zeros((B, 16), dtype=torch.float64, device=dev)
        x64 = torch.zeros((B, 3 * T), dtype=torch.float64, device=dev)
        g = (lambda k: _ptr(stage[k])) if stage is not None else (lambda k: None)
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            check(self._lib.npa_nrmp_stage(self._h, B, _ptr(nom_s), _ptr(nom_u), _ptr(ref_s), _ptr(ref_us), g("mu"),
                                           g("lam"), g("pts"), g("count"), _ptr(out_s), _ptr(out_u), _ptr(out_d),
                                           _ptr(info), _ptr(x64), C.c_void_p(stream)), "npa_nrmp_stage")
        torch.cuda.synchronize(dev)
        return dict(opt_s=out_s, opt_u=out_u, opt_d=out_d, info=info, x64=x64)

    def nrmp_params(self, nom_s, nom_u, stage=None):
        """The parameters the NRMP kernel builds before it solves (npa_nrmp_params): dict(A (B,T,3,3), B (B,T,3,2),
        C (B,T,3,1), fa (B,T,M,2), fb (B,T,M,1)) in the reference's shapes (robot.py:239-316, nrmp.py:220-261)."""
        T, M = self.T, self.nrmp_max_num
        nom_s = self._dev(nom_s)
        B = nom_s.shape[0]
        nom_u = self._dev(nom_u, (B, 2, T))
        dev = self.device
        abc = torch.zeros((B, T, 11), dtype=torch.float32, device=dev)
        f = torch.zeros((B, T, max(M, 1), 3), dtype=torch.float32, device=dev)
        g = (lambda k: _ptr(stage[k])) if stage is not None else (lambda k: None)
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            check(self._lib.npa_nrmp_params(self._h, B, _ptr(nom_s), _ptr(nom_u), g("mu"), g("lam"), g("pts"), g("count"),
                                            _ptr(abc), _ptr(f), C.c_void_p(stream)), "npa_nrmp_params")
        torch.cuda.synchronize(dev)
        a = abc.cpu().numpy()
        A = np.tile(np.eye(3, dtype=np.float32), (B, T, 1, 1))
        A[:, :, 0, 2], A[:, :, 1, 2] = a[:, :, 0], a[:, :, 1]
        Bm = a[:, :, 2:8].reshape(B, T, 3, 2)
        Cm = a[:, :, 8:11].reshape(B, T, 3, 1)
        fn = f.cpu().numpy()
        return dict(A=A, B=Bm, C=Cm, fa=fn[..., 0:2], fb=fn[..., 2:3])

    def nrmp_backward(self, nom_s, nom_u, ref_s, ref_us, stage, grad_s, grad_u, grad_d=None):
        """One NRMP solve plus dL/d(q_s[3], p_u, eta, d_max, d_min) per scene for upstream gradients
        (grad_s (B,3,T+1), grad_u (B,2,T), grad_d (B,1,T)|None).  Returns dict(opt_s, opt_u, opt_d,
        grad (B,8): q_s[0..2], p_u, eta, d_max, d_min, solver status; grad_nom_s (B,3,T+1): dL/d(the proximal
        centre), the upstream grad_s of the previous PAN iteration's solve)."""
        T = self.T
        nom_s = self._dev(nom_s)
        B = nom_s.shape[0]
        nom_u, ref_s, ref_us = self._dev(nom_u, (B, 2, T)), self._dev(ref_s, (B, 3, T + 1)), self._dev(ref_us, (B, T))
        gs, gu = self._dev(grad_s, (B, 3, T + 1)), self._dev(grad_u, (B, 2, T))
        gd = None if grad_d is None else self._dev(grad_d).reshape(B, T).contiguous()
        dev = self.device
        out_s = torch.empty((B, 3, T + 1), dtype=torch.float32, device=dev)
        out_u = torch.empty((B, 2, T), dtype=torch.float32, device=dev)
        out_d = torch.empty((B, 1, T), dtype=torch.float32, device=dev)
        gth = torch.zeros((B, 8), dtype=torch.float32, device=dev)
        gns = torch.zeros((B, 3, T + 1), dtype=torch.float32, device=dev)
        g = (lambda k: _ptr(stage[k])) if stage is not None else (lambda k: None)
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            check(self._lib.npa_nrmp_backward(self._h, B, _ptr(nom_s), _ptr(nom_u), _ptr(ref_s), _ptr(ref_us), g("mu"),
                                              g("lam"), g("pts"), g("count"), _ptr(out_s), _ptr(out_u), _ptr(out_d),
                                              _ptr(gs), _ptr(gu), _ptr(gd), _ptr(gth), _ptr(gns), None, C.c_void_p(stream)),
                  "npa_nrmp_backward")
        return dict(opt_s=out_s, opt_u=out_u, opt_d=out_d, grad=gth, grad_nom_s=gns)

    def forward_batch_grad(self, nom_s, nom_u, ref_s, ref_us, points=None, velocities=None, n_points=None):
        """`forward_batch` whose outputs are connected to `nrmp_layer.adjust_parameters` for autograd, as the
        reference's are through cvxpylayers (nrmp.py:79-95, :144; example/LON/LON_corridor.py:94-127):
        `loss(opt_s, opt_u, opt_d).backward()` fills `.grad` of q_s, p_u, eta, d_max, d_min (make them
        `requires_grad_(True)` leaves first).  The gradient follows the reference's autograd graph through the
        whole PAN loop: every executed NRMP solve contributes its direct dependence on the parameters (implicit
        differentiation of its KKT system, npa_nrmp_backward) and hands dL/d(its proximal centre) to the solve of
        the iteration before -- the one recurrent path the reference keeps (A/B/C, mu, lam are detached there:
        robot.py:272-316, dune.py:81, pan.py:207).  `recurrent=False` on the planner keeps only the last solve.
        Returns (opt_s (B,3,T+1), opt_u (B,2,T), opt_d (B,1,T)|None)."""
        f = self.nrmp_layer
        params = [f.q_s, f.p_u, f.eta, f.d_max, f.d_min]
        return _PanGrad.apply(self, (nom_s, nom_u, ref_s, ref_us, points, velocities, n_points), *params)

    def key_mode(self):
        """How the distance keys are computed for this checkpoint (npa_key_mode): dict(key_terms, measured_error, margin_e0)."""
        if self.no_obs:
            return dict(key_terms=0, measured_error=0.0, margin_e0=0.0)
        kt, er, e0 = C.c_int(), C.c_float(), C.c_float()
        check(self._lib.npa_key_mode(self._h, C.byref(kt), C.byref(er), C.byref(e0)), "npa_key_mode")
        return dict(key_terms=kt.value, measured_error=er.value, margin_e0=e0.value)

    def geo_report(self):
        """What npa_create measured about the geometric distance keys of this checkpoint (npa_geo_report)."""
        if self.no_obs:
            return None
        v = (C.c_float * 10)()
        check(self._lib.npa_geo_report(self._h, v, 10), "npa_geo_report")
        return dict(polygon_ok=bool(v[0]), measured_error=v[1], margin=v[2], refine_ratio=v[3], slope_estimate=v[4], g_far=v[5],
                    bf16_key_error=v[6], bf16_key_margin=v[7], table_key_error=v[8], table_key_margin=v[9])

    def audit(self, reset=False):
        """Run-time audit counters of the geometric-key margin (npa_audit_read; synchronises the device):
        dict(tiles, points, violations, worst_excess).  violations != 0 means the measured margin was exceeded by a real
        point: the kernel has switched itself to exact keys for every slice; rebuild the planner with NPA_KEY_TERMS=1."""
        if self.no_obs:
            return dict(tiles=0, points=0, violations=0, worst_excess=0.0)
        t, p, v, w = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_float()
        with torch.cuda.device(self.device):
            check(self._lib.npa_audit_read(self._h, C.byref(t), C.byref(p), C.byref(v), C.byref(w), 1 if reset else 0), "npa_audit_read")
        return dict(tiles=t.value, points=p.value, violations=v.value, worst_excess=w.value)

    def check_audit(self, fallback=False):
        """Poll the margin audit WITHOUT synchronising the device (npa_audit_peek: the selection kernel mirrors violations
        into pinned host memory).  Returns the violation count.  Non-zero: a real point exceeded the measured margin of the
        geometric keys -- the launch that saw it may have planned from a selection that missed a member, every launch
        since ran on exact keys (slow, right).  Warns once; fallback=True also switches the handle to network keys for
        good (npa_use_network_keys: buffers are re-made on the next forward_batch, prepared steps must be made again)."""
        if self.no_obs or not self._h.value:
            return 0
        v = C.c_uint64()
        check(self._lib.npa_audit_peek(self._h, C.byref(v)), "npa_audit_peek")
        if v.value:
            if not getattr(self, "_audit_warned", False):
                import warnings
                warnings.warn(f"neupan_amd: the geometric-key margin of this checkpoint was violated at run time ({v.value} "
                              "points): the selection kernel has switched to exact keys; plans of the launch that detected "
                              "it are suspect.  Call PAN.use_network_keys() or rebuild with NPA_KEY_TERMS=1.", RuntimeWarning)
                self._audit_warned = True
            if fallback:
                self.use_network_keys()
        return int(v.value)

    def use_network_keys(self):
        """Switch this planner from geometric to network keys for good (npa_use_network_keys)."""
        with torch.cuda.device(self.device):
            check(self._lib.npa_use_network_keys(self._h), "npa_use_network_keys")
        self._B = 0                                  # the workspace grows by the key buffer: re-made on the next call
        self._ws = self._state = None

    def selftest_flags(self):
        """What the create-time self-test changed: dict(warm_off, geo_rejected) (npa_selftest_flags)."""
        f = C.c_int()
        check(self._lib.npa_selftest_flags(self._h, C.byref(f)), "npa_selftest_flags")
        return dict(warm_off=bool(f.value & 1), geo_rejected=bool(f.value & 2))

    def last_qp_info(self):
        """(B,16) float64: per-scene diagnostics of the last QP solved by forward_batch
        (best iteration, merit, mu, status, iterations run)."""
        off = self._lib.npa_workspace_qp_info_offset(self._h, self._B)
        torch.cuda.synchronize(self.device)
        return self._ws[off:off + self._B * 16 * 8].view(torch.float64).reshape(self._B, 16).cpu().numpy()

    def profile(self, enable=True):
        check(self._lib.npa_profile_enable(self._h, 1 if enable else 0), "npa_profile_enable")

    def profile_read(self):
        a, s_, b, n = C.c_double(), C.c_double(), C.c_double(), C.c_int64()
        check(self._lib.npa_profile_read(self._h, C.byref(a), C.byref(s_), C.byref(b), C.byref(n)), "npa_profile_read")
        am, an = C.c_double(), C.c_int64()
        check(self._lib.npa_profile_read_aset(self._h, C.byref(am), C.byref(an)), "npa_profile_read_aset")
        return dict(dune_ms=a.value, select_ms=s_.value, nrmp_ms=b.value, launches=n.value, aset_ms=am.value, aset_launches=an.value)


class _PanGrad(torch.autograd.Function):
    """PAN loop forward on the HIP path; backward = npa_nrmp_backward on re-runs of the executed iterations, last
    to first, chained through the proximal centre."""

    @staticmethod
    def forward(ctx, pan, args, q_s, p_u, eta, d_max, d_min):
        nom_s, nom_u, ref_s, ref_us, points, velocities, n_points = args
        pan._push_adjust()
        pan.forward_begin(nom_s, nom_u, ref_s, ref_us, points, velocities, n_points)
        B, T = pan._B, pan.T
        # nominal trajectory each solve linearises around (cur_s, cur_u at the head of the workspace)
        wsf = pan._ws.view(torch.float32)
        n_s = B * 3 * (T + 1)
        off_u = (n_s + 3) // 4 * 4
        first = 0 if getattr(pan, "recurrent", True) else pan.iter_num - 1
        snaps = []
        use_rows = (not pan.no_obs) and points is not None
        vw = pan._workspace_views() if use_rows else None
        for k in range(pan.iter_num):
            if k >= first:
                snaps.append([k, wsf[:n_s].clone().reshape(B, 3, T + 1), wsf[off_u:off_u + B * 2 * T].clone().reshape(B, 2, T), None])
            pan.forward_iter(k)
            if k >= first and use_rows:
                # the sorted rows THIS iteration's QP was built from: the backward pass re-solves it from them instead of
                # re-running the DUNE stage (slice 0 is only refreshed in iteration 0 and stays valid: the QP never reads it)
                snaps[-1][3] = {key: vw[key].clone() for key in ("mu", "lam", "pts", "count")}
        out = pan.forward_end()
        ctx.pan = pan
        ctx.snaps = snaps
        ctx.iters = out["iters"].clone()
        ctx.args = (pan._dev(ref_s), pan._dev(ref_us), points, velocities, n_points)
        ctx.qs_shape = tuple(q_s.shape)
        ctx.no_obs = out["opt_d"] is None
        d = out["opt_d"] if out["opt_d"] is not None else torch.zeros((B, 1, T), device=pan.device)
        return out["opt_s"], out["opt_u"], d

    @staticmethod
    def backward(ctx, gs, gu, gd):
        pan = ctx.pan
        ref_s, ref_us, points, velocities, n_points = ctx.args
        B, T, dev = ctx.iters.shape[0], pan.T, pan.device
        gs = torch.zeros((B, 3, T + 1), device=dev) if gs is None else gs.contiguous()
        gu = torch.zeros((B, 2, T), device=dev) if gu is None else gu.contiguous()
        gd = None if (gd is None or ctx.no_obs) else gd.contiguous()
        tot = torch.zeros((B, 7), dtype=torch.float64, device=dev)
        for k, snap_s, snap_u, stage in reversed(ctx.snaps):
            ran = ctx.iters > k                        # scenes whose stop test ended the loop earlier skip this solve
            if not bool(ran.any()):
                continue
            r = pan.nrmp_backward(snap_s, snap_u, ref_s, ref_us, stage, gs, gu, gd)
            tot += torch.where(ran[:, None], r["grad"][:, :7].double(), torch.zeros_like(tot))
            m = ran[:, None, None]
            gs = torch.where(m, r["grad_nom_s"], gs)
            gu = torch.where(m, torch.zeros_like(gu), gu)
            if gd is not None:
                gd = torch.where(m, torch.zeros_like(gd), gd)
            if not bool((gs != 0).any()):
                break
        g = tot.sum(dim=0).float().cpu()
        gq = g[0:3].reshape(3, 1) if len(ctx.qs_shape) == 2 else g[0:3].sum().reshape(ctx.qs_shape)
        return None, None, gq, g[3].reshape(()), g[4].reshape(()), g[5].reshape(()), g[6].reshape(())


_STREAMS = {}


def forward_interleaved(planners, inputs, reset_state=False):
    """Plan several independent batches concurrently: `planners[i]` (one PAN per batch in flight, same configuration)
    plans `inputs[i]` (the positional arguments of forward_batch) on a stream of its own; the results are joined on the
    current stream.  Every kernel of the path is latency bound, so the chains of different batches fill each other's
    idle SIMDs.  reset_state: clear every planner's stop-criterion memory first (fresh planners), inside the staging
    launch.  Returns the list of output dicts (valid on the current stream)."""
    assert len(planners) == len(inputs) and len(planners) >= 1
    dev = planners[0].device
    cur = torch.cuda.current_stream(dev)
    key = (dev.index, len(planners))
    if key not in _STREAMS:
        _STREAMS[key] = [torch.cuda.Stream(device=dev) for _ in planners]
    streams = _STREAMS[key]
    outs = []
    for st, p, a in zip(streams, planners, inputs):
        st.wait_stream(cur)                      # inputs were produced on the current stream
        with torch.cuda.stream(st):
            outs.append(p.forward_batch(*a, reset_state=reset_state))
    for st, o in zip(streams, outs):
        cur.wait_stream(st)
        for t in o.values():                     # allocated on `st`, consumed on the current stream
            if isinstance(t, torch.Tensor):
                t.record_stream(cur)
    return outs


class StepGroup:
    """Steps prepared by PAN.make_step (one planner, one stream each), issued as ONE breadth-first library call
    (npa_forward_batch_group, include/neupan_amd.h): the staging launch of every step, then PAN iteration 0 of every
    step, ...  The launches and the results are those of calling the steps one after the other; every chain of the burst
    is running after the first 2 n launches instead of after 21 (n - 1).  `streams[i]` is the stream of steps[i].
    Members that share ONE stream (and a batch size and a configuration) run every stage as one MERGED launch over all their
    scenes (csrc/serve_group.hip: runs of <= 8 members; same results bitwise): 1 + 2K launches and one hardware queue for the
    whole run instead of per member.
    issue(n) enqueues the first n members (default: all) and returns their output dicts."""

    def __init__(self, steps, streams):
        from ._lib import NpaForwardCall
        calls = [getattr(s, "group_call", None) for s in steps]
        if any(c is None for c in calls):
            raise NeupanAmdError("StepGroup: every member must be a plain (non-graph) step of PAN.make_step")
        if len({c["flags"] for c in calls}) != 1 or len({str(c["device"]) for c in calls}) != 1:
            raise NeupanAmdError("StepGroup: the members must share the reset flag and the device")
        self.calls, self.flags, self.lib, self.device = calls, calls[0]["flags"], calls[0]["lib"], calls[0]["device"]
        self.steps, self.streams = list(steps), list(streams) if streams is not None else None
        self.arr = (NpaForwardCall * len(steps))()
        self._subs = {}                 # ctypes arrays of member selections other than a prefix (issue_members)
        names = [f[0] for f in NpaForwardCall._fields_]
        for a, c, st in zip(self.arr, calls, streams):
            h, B, n_stride, *rest = c["args"]
            vals = [h, B, n_stride, c["iter_num"], *rest, C.c_void_p(st.cuda_stream)]
            assert len(vals) == len(names)
            for k, v in zip(names, vals):
                setattr(a, k, v.value if isinstance(v, C.c_void_p) else v)

    def merged(self, n=None):
        """Would issue(n) run its members as merged launches (npa_forward_group_merged on runs of <= 8 members)?"""
        n = len(self.calls) if n is None else n
        want = (n + 7) // 8
        base, extra, lo = n // want, n % want, 0
        for r in range(want):
            ln = base + (1 if r < extra else 0)
            sub = (type(self.arr[0]) * ln)(*[self.arr[lo + i] for i in range(ln)])
            if ln < 2 or not self.lib.npa_forward_group_merged(ln, sub):
                return False
            lo += ln
        return True

    def issue(self, n=None):
        n = len(self.calls) if n is None else n
        if not 1 <= n <= len(self.calls):
            raise NeupanAmdError(f"StepGroup.issue: {n} members requested of {len(self.calls)}")
        return self._issue(self.arr, range(n))

    def issue_members(self, idx):
        """issue() for an arbitrary selection of members, in the order given (a partial round of a serving loop whose members
        are kept in chain-major order: neupan_amd.serve.StepLoop).  The library merges runs of consecutive members that share a
        stream exactly as for a prefix."""
        idx = tuple(int(i) for i in idx)
        if not idx or any(not 0 <= i < len(self.calls) for i in idx):
            raise NeupanAmdError(f"StepGroup.issue_members: members {idx} of {len(self.calls)}")
        if idx == tuple(range(len(idx))):
            return self._issue(self.arr, idx)
        sub = self._subs.get(idx)
        if sub is None:
            sub = self._subs[idx] = (type(self.arr[0]) * len(idx))(*[self.arr[i] for i in idx])
        return self._issue(sub, idx)

    def _issue(self, arr, members):
        n = len(members)
        # the group entry point's own argument checks, made here with a message (the library returns a bare NPA_E_ARG for them)
        hs = [arr[i].h for i in range(n)]
        if len(set(hs)) != n or any(not h for h in hs):
            raise NeupanAmdError("StepGroup.issue: the members of a group must be distinct live handles (one batch at a time per handle)")
        if any(arr[i].iter_num < 1 for i in range(n)):
            raise NeupanAmdError("StepGroup.issue: iter_num < 1")
        for m in members:
            self.calls[m]["validate"]()
        for m in members:
            pre = getattr(self.steps[m], "pre_issue", None)
            if pre is not None:
                with torch.cuda.stream(self.streams[m]):
                    pre()
        with torch.cuda.device(self.device):
            rc = self.lib.npa_forward_batch_group(n, arr, self.flags)
        if rc:
            check(rc, "npa_forward_batch_group")
        return [self.calls[m]["finish"]() for m in members]
