"""Engine-wide numeric mode.

``fp32`` (default) is the parity mode: every product runs on the exact f32 MFMA path and all
activations are fp32, so results track the reference's CPU fp32 PyTorch path to rounding.
``bf16`` is the throughput mode: bf16 activations / weight copies with fp32 accumulation,
fp32 master weights, fp32 cell state, statistics and loss lattice.

Select with ``edgedict_amd.set_compute_dtype('bf16')``, the ``EDGEDICT_DTYPE`` environment
variable, or per model via ``model.compute_dtype = torch.bfloat16``.
"""
import os

import torch

_NAMES = {"fp32": torch.float32, "float32": torch.float32, "f32": torch.float32,
          "bf16": torch.bfloat16, "bfloat16": torch.bfloat16}


def _parse(x):
    if isinstance(x, torch.dtype):
        if x not in (torch.float32, torch.bfloat16):
            raise ValueError("compute dtype must be float32 or bfloat16")
        return x
    try:
        return _NAMES[str(x).lower()]
    except KeyError:
        raise ValueError("unknown compute dtype %r (use 'fp32' or 'bf16')" % (x,))


# test hook: route bf16 recurrences through the generic step kernels instead of lstm_fast.hip
FORCE_GENERIC_LSTM = False

# bf16 encoders run as the layer-pipelined stack (csrc/encoder_stack.hip); False routes them
# through the per-layer kernels (test hook / A-B comparison)
USE_ENCODER_STACK = os.environ.get("EDGEDICT_ENCODER_STACK", "1") != "0"

# weight gradients that feed nothing downstream are accumulated straight into existing .grad
# buffers on the auxiliary stream, concurrently with the rest of the backward pass (side.py)
DEFER_WEIGHT_GRADS = os.environ.get("EDGEDICT_DEFER_DW", "1") != "0"
# the joint's output-bias gradient as partial column sums out of the loss-gradient kernel (csrc/rnnt_loss.hip rnnt_grad<T, true>)
# instead of a pass over the gradient matrix on the auxiliary stream
FUSED_DB2 = os.environ.get("EDGEDICT_FUSED_DB2", "1") != "0"

# the same for the per-layer LSTM blocks (fp32 mode, prediction network, small encoders): dW_ih / dW_hh / db of a layer
# run on the auxiliary stream under the BPTT of the layer below
DEFER_LSTM_WEIGHT_GRADS = os.environ.get("EDGEDICT_DEFER_LSTM_DW", "1") != "0"

# Transducer.forward runs the prediction network on the auxiliary stream, concurrently with the
# encoder (and, through autograd's stream replay, its backward concurrently with the encoder's)
DECODER_ON_AUX_STREAM = os.environ.get("EDGEDICT_DECODER_AUX", "1") != "0"

# ... enqueued before the encoder (1) or after it (0, default: measured 1.1 ms faster per step)
DECODER_ENQUEUE_FIRST = os.environ.get("EDGEDICT_DECODER_FIRST", "0") != "0"

# Transducer.forward(output_loss=True) with HOST-side lengths runs joint + loss on the packed
# lattice (only the cells inside each utterance's (T_b, U_b+1) box are materialised)
PACKED_LATTICE = os.environ.get("EDGEDICT_PACKED_LATTICE", "1") != "0"

# bf16 packed-lattice training path: log-softmax partials fused into the logits product's epilogue
# (csrc/gemm_nt256.hip) instead of a separate pass over the logits (rnnt_lse_gather)
FUSED_LSE = os.environ.get("EDGEDICT_FUSED_LSE", "1") != "0"

# bf16 inference on a chunk shorter than STACK_MIN_FRAMES (the streaming decoder): one native call with a fused launch
# per layer-frame instead of the per-layer kernels (EDGEDICT_STREAM_ENCODER_STEP=0 restores them)
STREAM_ENCODER_STEP = os.environ.get("EDGEDICT_STREAM_ENCODER_STEP", "1") != "0"
# ... for up to this many streams when the chunk is one or two encoder frames (the reference-native 75 ms chunk), and up
# to STREAM_STEP_MAX_ROWS streams for longer chunks - measured (tools/stream_bench.py, fused vs per-layer kernels, ms per
# chunk step through the module path): 1 frame, S = 64 / 256 / 1024: 0.34 / 0.33 / 0.64 vs 0.62 / 0.61 / 0.77; 4 frames,
# S = 64 / 256 / 1024: 0.84 / 0.94 / 2.20 vs 0.75 / 0.90 / 2.43 (the per-layer kernels batch the input product over the frames)
STREAM_STEP_MAX_ROWS_SHORT = int(os.environ.get("EDGEDICT_STREAM_STEP_MAX_ROWS_SHORT", "4096"))
# BatchedStreamDecoder: the native calls of a chunk step with pre-bound arguments and buffers (stream._ChunkPlan)
STREAM_FAST_CHUNK = os.environ.get("EDGEDICT_STREAM_FAST_CHUNK", "1") != "0"
STREAM_STEP_MAX_ROWS = int(os.environ.get("EDGEDICT_STREAM_STEP_MAX_ROWS", "16"))

# inputs shorter than this many frames use the per-layer path even in bf16 (see Encoder.forward)
STACK_MIN_FRAMES = int(os.environ.get("EDGEDICT_STACK_MIN_FRAMES", "24"))

_state = {"dtype": _parse(os.environ.get("EDGEDICT_DTYPE", "fp32")), "epoch": 0}


def set_compute_dtype(x):
    _state["dtype"] = _parse(x)


def get_compute_dtype():
    return _state["dtype"]


def bump_param_epoch():
    """Called by optimisers that update parameters behind autograd's back (raw kernels) so
    cached compute-dtype weight copies are refreshed."""
    _state["epoch"] += 1


def param_epoch():
    return _state["epoch"]
