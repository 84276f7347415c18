"""ctypes binding of libslam_engine.so (include/slam_engine.h).

PyTorch-ROCm is plumbing here: it owns device memory and streams; every compute call goes
through the C ABI with raw device pointers. There is NO CPU or eager fallback: if the HIP
library is missing the import of :func:`load_library` raises.
"""
from __future__ import annotations

import ctypes as C
import os
import re
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libslam_engine.so")
_lib = None

BUCKET_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int64, C.c_int64)


class SlamModelDesc(C.Structure):
    _fields_ = [
        ("n_layers", C.c_int32), ("hidden", C.c_int32), ("n_heads", C.c_int32), ("n_kv_heads", C.c_int32),
        ("head_dim", C.c_int32), ("intermediate", C.c_int32), ("vocab", C.c_int32), ("pad_token_id", C.c_int32),
        ("rms_eps", C.c_float), ("rope_theta", C.c_float),
    ]


class SlamTensorInfo(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("offset", C.c_int64), ("rows", C.c_int64), ("cols", C.c_int64)]


def header_symbols() -> List[str]:
    """Every function name declared in include/slam_engine.h (used by the export test)."""
    hdr = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "slam_engine.h")
    txt = open(hdr).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(slam_[a-z0-9_]+)\s*\(", txt)) - {"slam_bucket_cb"})


def load_library(path: Optional[str] = None):
    """Load the HIP engine; raises OSError (loudly) when it has not been built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("SLAM_ENGINE_LIB", _LIB_PATH)
    # PyTorch-ROCm first: its wheel carries its own libamdhip64; the engine library must bind to THAT runtime instance (the one
    # that owns the device memory and streams it is handed), not to a second copy loaded from /opt/rocm before torch came up
    # (kernel launches then fail with hipErrorNoDevice)
    import torch  # noqa: F401
    if not os.path.exists(p):
        raise OSError(f"{p} not found: build it with `python -m slamkit_amd.csrc.build` "
                      f"(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(p)
    vp, i32, i64, f32, f64, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_double, C.c_size_t
    sig = {
        "slam_engine_create": (C.c_int, [C.POINTER(SlamModelDesc), C.POINTER(vp)]),
        "slam_engine_destroy": (None, [vp]),
        "slam_last_error": (C.c_char_p, [vp]),
        "slam_version": (C.c_char_p, []),
        "slam_param_count": (i64, [vp]),
        "slam_tensor_count": (i32, [vp]),
        "slam_tensor_info": (C.c_int, [vp, i32, C.POINTER(SlamTensorInfo)]),
        "slam_bind_params": (C.c_int, [vp, vp, vp]),
        "slam_bind_params_t": (C.c_int, [vp, vp]),
        "slam_refresh_transposed": (C.c_int, [vp, vp]),
        "slam_workspace_bytes": (sz, [vp, i64]),
        "slam_bind_workspace": (C.c_int, [vp, vp, sz, i64]),
        "slam_forward": (C.c_int, [vp, vp, vp, vp, vp, vp, i32, i32, f64, vp, vp, vp]),
        "slam_backward": (C.c_int, [vp, f32, i32, BUCKET_CB, vp, vp]),
        "slam_bucket_stream": (vp, [vp]),
        "slam_set_logit_mask": (C.c_int, [vp, vp]),
        "slam_padded_vocab": (i32, [vp]),
        "slam_seq_loglik": (C.c_int, [vp, vp, i32, i32, vp, vp, vp]),
        "slam_scale_loss_rows": (C.c_int, [vp, vp, i32, i32, vp]),
        "slam_grad_norm": (C.c_int, [vp, f32, vp, vp]),
        "slam_adamw_step": (C.c_int, [vp, vp, vp, vp, vp, f64, f64, f64, f64, f64, i32, i32, vp]),
        "slam_adamw_step_bf16": (C.c_int, [vp, vp, vp, vp, f64, f64, f64, f64, f64, i32, i32, vp]),
        "slam_adamw_step_bf16_moments": (C.c_int, [vp, vp, vp, vp, vp, f64, f64, f64, f64, f64, i32, i32, vp]),
        "slam_adamw_range_bf16_moments": (C.c_int, [vp, i64, i64, vp, vp, vp, vp, f64, f64, f64, f64, f64, i32, i32, vp]),
        "slam_grad_chunk_elems": (i64, []),
        "slam_grad_sumsq_chunks": (C.c_int, [vp, i64, i64, vp, vp]),
        "slam_grad_norm_from_chunks": (C.c_int, [vp, vp, f32, vp, vp]),
        "slam_adamw_range": (C.c_int, [vp, i64, i64, vp, vp, vp, vp, f64, f64, f64, f64, f64, i32, i32, vp]),
        "slam_adamw_range_bf16": (C.c_int, [vp, i64, i64, vp, vp, vp, f64, f64, f64, f64, f64, i32, i32, vp]),
        "slam_add_param_wait": (C.c_int, [vp, i64, i64, vp]),
        "slam_param_wait_ms": (C.c_int, [vp, C.POINTER(C.c_float)]),
        "slam_param_wait_untimed": (C.c_int, [vp, C.POINTER(C.c_int64)]),
        "slam_comm_unique_id": (C.c_int, [vp, C.c_int32]),
        "slam_comm_init": (C.c_int, [vp, vp, C.c_int32, C.c_int32]),
        "slam_comm_destroy": (C.c_int, [vp]),
        "slam_allreduce_grads_async": (C.c_int, [vp, i64, i64, C.c_int32, vp]),
        "slam_comm_finish": (C.c_int, [vp, vp]),
        "slam_reduce_scatter_grads_async": (C.c_int, [vp, i64, i64, C.c_int32, vp]),
        "slam_allgather_params_async": (C.c_int, [vp, i64, i64, vp]),
        "slam_gateup_launch_ms": (C.c_int, [vp, C.POINTER(C.c_float), C.c_int32]),
        "slam_family_ms": (C.c_int, [vp, C.POINTER(C.c_int32), C.POINTER(C.c_float), C.c_int32, C.POINTER(C.c_int32)]),
        "slam_family_name": (C.c_char_p, [C.c_int32]),
        "slam_pack_grads_bf16": (C.c_int, [vp, i64, i64, vp, vp]),
        "slam_unpack_grads_bf16": (C.c_int, [vp, i64, i64, vp, vp]),
        "slam_set_grad_image": (C.c_int, [vp, vp]),
        "slam_join": (C.c_int, [vp, vp]),
        "slam_zero_grads": (C.c_int, [vp, vp]),
        "slam_cast_params": (C.c_int, [vp, vp, vp]),
        "slam_set_option": (C.c_int, [vp, C.c_char_p, i64]),
        "slam_op_gemm_nt": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
        "slam_op_gemm_nt_swiglu": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp]),
        "slam_op_gemm_nt_dswiglu": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, vp]),
        "slam_op_gemm_nn": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp]),
        "slam_op_gemm_tn_workspace": (sz, [C.c_int, C.c_int, C.c_int]),
        "slam_op_gemm_tn": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
        "slam_op_gemm_tn_image": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp]),
        "slam_op_rmsnorm_fwd": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, f32, vp]),
        "slam_op_rmsnorm_bwd_workspace": (sz, [C.c_int, C.c_int]),
        "slam_op_rmsnorm_bwd": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, vp]),
        "slam_op_rope": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, f32, C.c_int, vp, vp]),
        "slam_op_swiglu_fwd": (C.c_int, [vp, vp, C.c_int, C.c_int, vp]),
        "slam_op_swiglu_bwd": (C.c_int, [vp, vp, C.c_int, C.c_int, vp]),
        "slam_op_attn_fwd": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
        "slam_op_attn_bwd_workspace": (sz, [C.c_int, C.c_int, C.c_int]),
        "slam_op_attn_bwd": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
        "slam_op_cross_entropy": (C.c_int, [vp, vp, f64, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
        "slam_op_embed_bwd_workspace": (sz, [C.c_int, C.c_int]),
        "slam_op_embed_bwd": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    }
    for name, (res, args) in sig.items():
        if path is not None and not hasattr(lib, name):
            continue  # an explicitly named OTHER build (A/B tooling against an older library): bind what it has
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    lib._slam_signatures = sig
    if path is None:
        _lib = lib
    return lib


def _ptr(t) -> Optional[int]:
    return None if t is None else int(t.data_ptr())


def current_stream_ptr() -> int:
    import torch
    return int(torch.cuda.current_stream().cuda_stream)


class EngineError(RuntimeError):
    pass


@dataclass
class TensorSpec:
    name: str
    offset: int
    rows: int
    cols: int

    @property
    def numel(self) -> int:
        return self.rows * self.cols


class Engine:
    """Thin owner of a SlamEngine handle plus the torch tensors it borrows."""

    def __init__(self, desc: SlamModelDesc):
        self.lib = load_library()
        self.desc = desc
        h = C.c_void_p()
        rc = self.lib.slam_engine_create(C.byref(desc), C.byref(h))
        if rc != 0:
            raise EngineError(f"slam_engine_create failed ({rc}): unsupported model description")
        self.h = h
        self.n_params = int(self.lib.slam_param_count(h))
        self.tensors: Dict[str, TensorSpec] = {}
        info = SlamTensorInfo()
        for i in range(self.lib.slam_tensor_count(h)):
            self._ck(self.lib.slam_tensor_info(h, i, C.byref(info)))
            self.tensors[info.name.decode()] = TensorSpec(info.name.decode(), info.offset, info.rows, info.cols)
        self._keep = {}

    def _ck(self, rc: int):
        if rc != 0:
            msg = self.lib.slam_last_error(self.h)
            raise EngineError(f"engine call failed ({rc}): {msg.decode() if msg else ''}")

    def close(self):
        if getattr(self, "h", None):
            self.lib.slam_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- binding ------------------------------------------------------------------------------
    def bind_params(self, params_bf16, grads_f32=None):
        import torch
        assert params_bf16.dtype == torch.bfloat16 and params_bf16.numel() == self.n_params and params_bf16.is_cuda
        if grads_f32 is not None:
            assert grads_f32.dtype == torch.float32 and grads_f32.numel() == self.n_params
        self._keep["params"], self._keep["grads"] = params_bf16, grads_f32
        self._ck(self.lib.slam_bind_params(self.h, _ptr(params_bf16), _ptr(grads_f32)))

    def bind_params_t(self, params_t_bf16):
        self._keep["params_t"] = params_t_bf16
        self._ck(self.lib.slam_bind_params_t(self.h, _ptr(params_t_bf16)))

    def refresh_transposed(self, stream=None):
        self._ck(self.lib.slam_refresh_transposed(self.h, stream if stream is not None else current_stream_ptr()))

    def workspace_bytes(self, max_tokens: int) -> int:
        return int(self.lib.slam_workspace_bytes(self.h, max_tokens))

    def bind_workspace(self, ws, max_tokens: int):
        self._keep["ws"] = ws
        self._ck(self.lib.slam_bind_workspace(self.h, _ptr(ws), ws.numel() * ws.element_size(), max_tokens))

    def set_option(self, key: str, value: int):
        self._ck(self.lib.slam_set_option(self.h, key.encode(), int(value)))

    # -- step ---------------------------------------------------------------------------------
    def forward(self, ids, labels=None, position_ids=None, seg_start=None, seg_end=None, B=1, T=1,
                num_items: float = 0.0, loss_out=None, logits_out=None, stream: Optional[int] = None):
        self._ck(self.lib.slam_forward(self.h, _ptr(ids), _ptr(labels), _ptr(position_ids), _ptr(seg_start),
                                       _ptr(seg_end), B, T, float(num_items), _ptr(loss_out), _ptr(logits_out),
                                       stream if stream is not None else current_stream_ptr()))

    def backward(self, grad_scale: float = 1.0, bucket_layers: int = 0,
                 bucket_cb: Optional[Callable[[int, int], None]] = None, stream: Optional[int] = None, final: int = 0):
        """final (option "grad_final_next"): 1 = last backward of its optimizer step (the final-value stores emit the
        gradient-norm partials), 2 = the same with the final values kept in bf16 only, in the set_grad_image buffer."""
        if final:
            self.set_option("grad_final_next", int(final))
        if bucket_cb is None:
            cb = C.cast(None, BUCKET_CB)
        else:
            # third argument: the stream the range is complete on (slam_bucket_stream; None = the backward stream)
            cb = BUCKET_CB(lambda _u, off, cnt: bucket_cb(int(off), int(cnt), self.lib.slam_bucket_stream(self.h)))
        self._ck(self.lib.slam_backward(self.h, float(grad_scale), int(bucket_layers), cb, None,
                                        stream if stream is not None else current_stream_ptr()))

    def set_logit_mask(self, mask_u8=None):
        """mask_u8: uint8 device tensor of padded_vocab() bytes (non-zero = column outside the softmax) or None."""
        self._ck(self.lib.slam_set_logit_mask(self.h, _ptr(mask_u8) if mask_u8 is not None else None))

    def padded_vocab(self) -> int:
        return int(self.lib.slam_padded_vocab(self.h))

    def seq_loglik(self, labels, B, T, ll_out, cnt_out, stream=None):
        self._ck(self.lib.slam_seq_loglik(self.h, _ptr(labels), B, T, _ptr(ll_out), _ptr(cnt_out),
                                          stream if stream is not None else current_stream_ptr()))

    def scale_loss_rows(self, seq_coef, B, T, stream=None):
        self._ck(self.lib.slam_scale_loss_rows(self.h, _ptr(seq_coef), B, T,
                                               stream if stream is not None else current_stream_ptr()))

    def grad_norm(self, max_norm: float, norm_out, stream=None):
        self._ck(self.lib.slam_grad_norm(self.h, float(max_norm), _ptr(norm_out),
                                         stream if stream is not None else current_stream_ptr()))

    def adamw_step(self, master, exp_avg, exp_avg_sq, norm_out, lr, beta1, beta2, eps, weight_decay, step,
                   zero_grad=True, stream=None):
        import torch
        if exp_avg.dtype == torch.bfloat16:  # fp32 master + bf16 moments
            self._ck(self.lib.slam_adamw_step_bf16_moments(self.h, _ptr(master), _ptr(exp_avg), _ptr(exp_avg_sq), _ptr(norm_out),
                                                           float(lr), float(beta1), float(beta2), float(eps), float(weight_decay),
                                                           int(step), int(bool(zero_grad)),
                                                           stream if stream is not None else current_stream_ptr()))
            return
        self._ck(self.lib.slam_adamw_step(self.h, _ptr(master), _ptr(exp_avg), _ptr(exp_avg_sq), _ptr(norm_out),
                                          float(lr), float(beta1), float(beta2), float(eps), float(weight_decay),
                                          int(step), int(bool(zero_grad)),
                                          stream if stream is not None else current_stream_ptr()))

    def adamw_step_bf16(self, exp_avg, exp_avg_sq, norm_out, lr, beta1, beta2, eps, weight_decay, step, zero_grad=True,
                        stream=None):
        """bf16 parameters + bf16 moments, updated in place (the recipe's optimizer precision)."""
        import torch
        assert exp_avg.dtype == torch.bfloat16 and exp_avg_sq.dtype == torch.bfloat16
        self._ck(self.lib.slam_adamw_step_bf16(self.h, _ptr(exp_avg), _ptr(exp_avg_sq), _ptr(norm_out), float(lr),
                                               float(beta1), float(beta2), float(eps), float(weight_decay), int(step),
                                               int(bool(zero_grad)),
                                               stream if stream is not None else current_stream_ptr()))

    # -- sharded optimizer step (data-parallel "rs_ag") ----------------------------------------------------
    def grad_chunk_info(self):
        """(elements per gradient-norm chunk, number of chunks of the flat buffer)."""
        c = int(self.lib.slam_grad_chunk_elems())
        return c, (self.n_params + c - 1) // c

    def grad_sumsq_chunks(self, offset: int, count: int, chunk_sums, stream=None):
        self._ck(self.lib.slam_grad_sumsq_chunks(self.h, int(offset), int(count), _ptr(chunk_sums),
                                                 stream if stream is not None else current_stream_ptr()))

    def grad_norm_from_chunks(self, chunk_sums, max_norm: float, norm_out, stream=None):
        self._ck(self.lib.slam_grad_norm_from_chunks(self.h, _ptr(chunk_sums), float(max_norm), _ptr(norm_out),
                                                     stream if stream is not None else current_stream_ptr()))

    def adamw_range(self, offset: int, count: int, master, exp_avg, exp_avg_sq, norm_out, lr, beta1, beta2, eps, weight_decay,
                    step, zero_grad=False, stream=None):
        """AdamW on elements [offset, offset + count): master / exp_avg / exp_avg_sq are FULL-SIZE flat tensors here (the
        range's slices are passed down); master = None selects the bf16-state form."""
        import torch
        o, n = int(offset), int(count)
        st = stream if stream is not None else current_stream_ptr()
        if master is None:
            self._ck(self.lib.slam_adamw_range_bf16(self.h, o, n, _ptr(exp_avg[o:o + n]), _ptr(exp_avg_sq[o:o + n]), _ptr(norm_out),
                                                    float(lr), float(beta1), float(beta2), float(eps), float(weight_decay),
                                                    int(step), int(bool(zero_grad)), st))
        elif exp_avg.dtype == torch.bfloat16:
            self._ck(self.lib.slam_adamw_range_bf16_moments(self.h, o, n, _ptr(master[o:o + n]), _ptr(exp_avg[o:o + n]),
                                                            _ptr(exp_avg_sq[o:o + n]), _ptr(norm_out), float(lr), float(beta1),
                                                            float(beta2), float(eps), float(weight_decay), int(step),
                                                            int(bool(zero_grad)), st))
        else:
            self._ck(self.lib.slam_adamw_range(self.h, o, n, _ptr(master[o:o + n]), _ptr(exp_avg[o:o + n]), _ptr(exp_avg_sq[o:o + n]),
                                               _ptr(norm_out), float(lr), float(beta1), float(beta2), float(eps),
                                               float(weight_decay), int(step), int(bool(zero_grad)), st))

    def add_param_wait(self, offset: int, count: int, event):
        """event: a recorded torch.cuda.Event; kept alive here until the next forward consumed it."""
        self._keep.setdefault("param_events", []).append(event)
        if len(self._keep["param_events"]) > 256:
            del self._keep["param_events"][:128]
        self._ck(self.lib.slam_add_param_wait(self.h, int(offset), int(count), C.c_void_p(int(event.cuda_event))))

    def param_wait_ms(self) -> float:
        out = C.c_float(0.0)
        self._ck(self.lib.slam_param_wait_ms(self.h, C.byref(out)))
        return float(out.value)

    def param_wait_untimed(self) -> int:
        """Parameter waits since the last call that the engine could not time (0 = param_wait_ms() was complete)."""
        out = C.c_int64(0)
        self._ck(self.lib.slam_param_wait_untimed(self.h, C.byref(out)))
        return int(out.value)

    def gateup_launch_ms(self, n_layers: int):
        """Durations (ms) of the gate|up projection launches of the last forward (option time_gateup = 1)."""
        out = (C.c_float * n_layers)()
        self._ck(self.lib.slam_gateup_launch_ms(self.h, out, n_layers))
        return [float(v) for v in out]

    def family_ms(self, capacity: int = 4096):
        """[(family name, ms)] of the last forward + backward in launch order (option time_families = 1)."""
        fam = (C.c_int32 * capacity)()
        ms = (C.c_float * capacity)()
        n = C.c_int32(0)
        self._ck(self.lib.slam_family_ms(self.h, fam, ms, capacity, C.byref(n)))
        return [(self.lib.slam_family_name(int(fam[i])).decode(), float(ms[i])) for i in range(n.value)]

    def pack_grads_bf16(self, offset: int, count: int, dst_bf16, stream=None):
        """dst_bf16[0:count] = bf16(grads[offset:offset+count]); dst_bf16: a bf16 device tensor (view) of >= count elements."""
        self._ck(self.lib.slam_pack_grads_bf16(self.h, int(offset), int(count), _ptr(dst_bf16),
                                               stream if stream is not None else current_stream_ptr()))

    def set_grad_image(self, grads_bf16):
        """The next backward also writes every final gradient value into `grads_bf16` (bf16, n_params elements; None = off)."""
        if grads_bf16 is not None:
            import torch
            assert grads_bf16.dtype == torch.bfloat16 and grads_bf16.numel() == self.n_params and grads_bf16.is_cuda
        self._keep["grad_image"] = grads_bf16
        self._ck(self.lib.slam_set_grad_image(self.h, _ptr(grads_bf16)))

    def unpack_grads_bf16(self, offset: int, count: int, src_bf16, stream=None):
        self._ck(self.lib.slam_unpack_grads_bf16(self.h, int(offset), int(count), _ptr(src_bf16),
                                                 stream if stream is not None else current_stream_ptr()))

    # ---- engine-side gradient exchange (RCCL looked up at run time; include/slam_engine.h slam_comm_*) --------------------------
    @staticmethod
    def comm_unique_id() -> bytes:
        """128 bytes from ncclGetUniqueId (rank 0 calls this and ships them to every rank)."""
        buf = C.create_string_buffer(128)
        rc = load_library().slam_comm_unique_id(buf, 128)
        if rc != 0:
            raise RuntimeError(f"slam_comm_unique_id failed with {rc}" + (" (librccl.so.1 not found)" if rc == -4 else ""))
        return buf.raw

    def comm_init(self, unique_id: bytes, rank: int, world: int):
        self._ck(self.lib.slam_comm_init(self.h, C.create_string_buffer(unique_id, 128), int(rank), int(world)))

    def comm_destroy(self):
        self._ck(self.lib.slam_comm_destroy(self.h))

    def allreduce_grads_async(self, offset: int, count: int, bf16_exchange: bool = False, ready_stream=None):
        """Sum grads[offset:offset+count] over the communicator on the engine's communication stream, after the work
        enqueued so far on `ready_stream` (raw handle; None = torch's current stream)."""
        self._ck(self.lib.slam_allreduce_grads_async(self.h, int(offset), int(count), int(bool(bf16_exchange)),
                                                     ready_stream if ready_stream else current_stream_ptr()))

    def reduce_scatter_grads_async(self, offset: int, count: int, bf16_exchange: bool = False, ready_stream=None):
        """Reduce-scatter grads[offset:offset+count] over the communicator: this rank ends up with the summed shard
        [offset + rank * count / world, ...) (engine communication stream, behind `ready_stream`)."""
        self._ck(self.lib.slam_reduce_scatter_grads_async(self.h, int(offset), int(count), int(bool(bf16_exchange)),
                                                          ready_stream if ready_stream else current_stream_ptr()))

    def allgather_params_async(self, offset: int, count: int, ready_stream=None):
        """All-gather the bf16 parameters of the bucket from their owners; the next forward waits per bucket."""
        self._ck(self.lib.slam_allgather_params_async(self.h, int(offset), int(count),
                                                      ready_stream if ready_stream else current_stream_ptr()))

    def comm_finish(self, stream=None):
        self._ck(self.lib.slam_comm_finish(self.h, stream if stream is not None else current_stream_ptr()))

    def zero_grads(self, stream=None):
        self._ck(self.lib.slam_zero_grads(self.h, stream if stream is not None else current_stream_ptr()))

    def join(self, stream=None):
        """Order `stream` after a pending overlapped optimizer step (call before reading the flat buffers with torch)."""
        self._ck(self.lib.slam_join(self.h, stream if stream is not None else current_stream_ptr()))

    def cast_params(self, master, stream=None):
        self._ck(self.lib.slam_cast_params(self.h, _ptr(master),
                                           stream if stream is not None else current_stream_ptr()))
