"""Renderer: ray chunk loop + reshape to rays.restore_shape (reference models/renderer.py:7-65)."""
import os

import torch
import torch.nn as nn

# read once.  NVFI_EVAL_CHUNK: upper bound of the internal test-mode chunk (rays); NVFI_EVAL_STREAMS: 1 = the reference's plain
# loop on the current stream; NVFI_EVAL_WS_BYTES: byte budget for the chunks in flight (workspace + the (R, S) weights output of each)
_EVAL_CHUNK_MAX = int(os.environ.get("NVFI_EVAL_CHUNK", "32768"))
_EVAL_STREAMS = 1 if os.environ.get("NVFI_EVAL_STREAMS", "2") == "1" else 2
_EVAL_WS_BYTES = int(float(os.environ.get("NVFI_EVAL_WS_BYTES", str(8 << 30))))


class Renderer(nn.Module):
    def __init__(self, tensorf, batch_size, test_batch_size, ray_chunk, distance_scale=1, lindisp=False,
                 perturb=True, tensorf_sample=True, ndc=False):
        super().__init__()
        self.tensorf = tensorf
        self.batch_size, self.test_batch_size = batch_size, test_batch_size
        self.lindisp, self.perturb, self.distance_scale = lindisp, perturb, distance_scale
        self.tensorf_sample, self.ndc, self.ray_chunk = tensorf_sample, ndc, ray_chunk
        self.eval_chunk = None          # None: derived from the byte budget below; an int pins the internal test-mode chunk (rays)
        self.eval_ws_bytes = _EVAL_WS_BYTES

    def _eval_chunk(self, t, transfer_vel):
        """Internal chunk of a test-mode frame: the largest multiple of the caller's ray_chunk, at most NVFI_EVAL_CHUNK, whose in-flight
        memory - _EVAL_STREAMS x (nvfi_render_workspace_bytes_t + the (R, S) weights) - stays inside `eval_ws_bytes` (8 GiB by default; the
        workspace and weights grow with nSamples, ~1000 after the last upsampling).  A caller who lowered ray_chunk to bound memory keeps that
        bound by lowering eval_ws_bytes (or pinning eval_chunk): the chunk never drops below ray_chunk, which is the reference's loop."""
        base = max(1, int(self.ray_chunk))
        if self.eval_chunk is not None:
            return max(1, int(self.eval_chunk))
        field = getattr(self.tensorf, "nvfi", None)
        if field is None or not hasattr(field, "render_workspace_bytes"):
            return base
        key = (base, int(field.nSamples), self.eval_ws_bytes, bool(transfer_vel), float(t))
        c = self.__dict__.get("_eval_chunk_cache")
        if c is not None and c[0] == key:
            return c[1]
        chunk = max(base, _EVAL_CHUNK_MAX // base * base)
        while chunk > base:
            need = _EVAL_STREAMS * (field.render_workspace_bytes(chunk, t, transfer=transfer_vel) + chunk * (int(field.nSamples) + 8) * 4)
            if need <= self.eval_ws_bytes:
                break
            chunk = max(base, (chunk // 2) // base * base)
        self.__dict__["_eval_chunk_cache"] = (key, chunk)
        return chunk

    def forward(self, t, rays, white_background=False, transfer_vel=False):
        ray_o = rays.ray_origins.reshape(-1, 3)
        ray_d = rays.ray_directions.reshape(-1, 3)
        n_all = ray_o.shape[0]
        outs = [[], [], [], [], []]
        chunk = self.ray_chunk
        if ray_o.is_cuda and not torch.is_grad_enabled() and not self.tensorf.training:
            # test-mode rays carry no jitter and are independent (tests/test_gpu_edges.py: every prefix of a render equals the render), so the
            # reference's ray_chunk - a memory bound for its (R, S, .) torch intermediates - need not be the launch granularity here: a frame
            # goes through in pieces of up to NVFI_EVAL_CHUNK rays (default 32768: 16 MB of weights at 128 samples per ray; bounded by a byte
            # budget, _eval_chunk), 16x fewer launches
            chunk = self._eval_chunk(t, transfer_vel)
        n_chunks = n_all // chunk + int(n_all % chunk > 0)
        fn = self.tensorf.render_ray_transfer if transfer_vel else self.tensorf.render_ray
        # a test-mode frame is some hundred independent chunks: issued alternately on two side streams, one chunk's velocity warp (matrix pipe)
        # runs beside the other's plane gathers (HBM) - every chunk call owns its workspace and outputs, nothing is shared but the weights.
        # NVFI_EVAL_STREAMS=1: the reference's plain loop on the current stream
        side = None
        if n_chunks >= 4 and ray_o.is_cuda and not torch.is_grad_enabled() and _EVAL_STREAMS != 1:
            side = self.__dict__.get("_eval_streams")
            if side is None or side[0].device != ray_o.device:
                side = self.__dict__["_eval_streams"] = [torch.cuda.Stream(device=ray_o.device) for _ in range(2)]
            main = torch.cuda.current_stream(ray_o.device)
            for s_ in side:
                s_.wait_stream(main)
        for c in range(n_chunks):
            r_o = ray_o[c * chunk:(c + 1) * chunk]
            r_d = ray_d[c * chunk:(c + 1) * chunk]
            if side is None:
                res = fn(t, r_o, r_d, white_background, self.ndc)
            else:
                with torch.cuda.stream(side[c & 1]):
                    res = fn(t, r_o, r_d, white_background, self.ndc)
                for v in res:
                    if torch.is_tensor(v):
                        v.record_stream(main)          # allocated on the side stream, consumed (torch.cat below, the caller) on the current one
            for lst, v in zip(outs, res):
                lst.append(v)
        if side is not None:
            for s_ in side:
                main.wait_stream(s_)
        # (one chunk - every training batch - needs no concatenation: five copy launches less per render)
        rgb_map, depth_map, acc_map, weights, extra = [o[0] if len(o) == 1 else torch.cat(o, 0) for o in outs]
        shp = tuple(rays.restore_shape)
        return (rgb_map.reshape(*shp, 3), depth_map.reshape(*shp), acc_map.reshape(*shp),
                weights.reshape(*shp, -1), extra.reshape(*shp, extra.shape[-1]))

    def render(self, t, rays, white_background=False, mode="train", transfer_vel=False):
        if mode == "train":
            if not self.tensorf.training:     # (the reference calls .train()/.eval() unconditionally; the recursion is pure host cost)
                self.tensorf.train()
            return self.forward(t, rays, white_background)
        if self.tensorf.training:
            self.tensorf.eval()
        with torch.no_grad():
            return self.forward(t, rays, white_background, transfer_vel=transfer_vel)
