"""tf_util -- host-side mirror of the reference's layer wrappers (utils/tf_util.py:52,120,327,512,594),
inference only, on torch.

The reference builds TF variables under ``tf.variable_scope``; here a ``VariableStore`` plays that role: a
layer looks its variables up by scope path (``layer1/conv_kv/weights`` ...) and creates them on first use from
a seed, so a model is defined by calling the same functions in the same order as the reference.  Variable
names and shapes follow the reference (weights [kh,kw,cin,cout] flattened to [kh*kw*cin, cout], biases,
bn/gamma, bn/beta, bn/moving_mean, bn/moving_variance).

These are plain GEMMs and go to the vendor BLAS through torch (they are NOT custom ops in the reference
either).  Inference batch norm (tf.contrib.layers.batch_norm, epsilon 1e-3) is folded into the GEMM:
    W' = W * s,  b' = (b - mean) * s + beta,  s = gamma / sqrt(var + 1e-3)
"""
import contextlib
import hashlib
import math

import numpy as np
import torch

from pointasnl_amd import _hip

BN_EPS = 1e-3
FUSE_RELU_EPILOGUE = True
WEIGHTS_TRANSPOSED_MIN_K = 2048  # dense layers contracting at least this many inputs keep their weights transposed (see _dense)


class VariableStore:
    """Seeded variable container + scope stack (the stand-in for TF's graph collections)."""

    def __init__(self, seed=0, device="cuda", randomize_bn=False):
        self.seed = int(seed)
        self.device = device
        self.randomize_bn = randomize_bn  # non-trivial gamma/beta/mean/var so that BN folding is actually tested
        self.vars = {}
        self.decays = {}  # weights variable -> wd: the reference's 'losses' collection (tf_util.py:30-49: wd * l2_loss(var))
        self._folded = {}
        self._scope = []

    # ---- scopes
    @contextlib.contextmanager
    def scope(self, name):
        self._scope.append(name)
        try:
            yield
        finally:
            self._scope.pop()

    def path(self, name):
        return "/".join(self._scope + [name])

    # ---- variables
    def _rng(self, full):
        h = hashlib.sha256(f"{self.seed}:{full}".encode()).digest()
        return np.random.Generator(np.random.PCG64(int.from_bytes(h[:8], "little")))

    def load(self, variables, strict=True):
        """Take variable VALUES from a mapping keyed by the reference's TF variable names -- a trained checkpoint
        exported to npz, or any {name: array}:

            reader = tf.train.load_checkpoint(ckpt)          # on a machine that has TF 1.x
            np.savez("pointasnl.npz", **{n: reader.get_tensor(n) for n in reader.get_variable_to_shape_map()})
            store.load(np.load("pointasnl.npz"))

        Names are the graph's own ('layer1/layer1/conv_kv/weights', '.../bn/moving_variance', ...; a ':0' suffix and
        optimizer slots are ignored).  Conv kernels [kh,kw,cin,cout] / [k,cin,cout] are flattened to [kh*kw*cin, cout]
        (row-major: the [1,W] VALID kernels become one GEMM over the (W,C) window).  Shapes are validated when a layer
        asks for its variables.  strict: a layer that asks for a variable the mapping does not hold raises instead of
        drawing a seeded one.  Every cached BN-folded / concatenated weight is dropped."""
        for name in getattr(variables, "files", None) or variables.keys():
            leaf = name.split(":")[0].split("/")[-1]
            if leaf not in ("weights", "biases", "beta", "gamma", "moving_mean", "moving_variance"):
                continue  # Adam slots, global_step, batch counters
            a = np.asarray(variables[name], dtype=np.float32)
            if leaf == "weights" and a.ndim > 2:
                a = a.reshape(-1, a.shape[-1])
            self.vars[name.split(":")[0]] = torch.tensor(a, dtype=torch.float32, device=self.device)
        self.strict = bool(strict)
        self._folded.clear()
        return self

    def assign(self, full_name, value):
        """Overwrite one variable (2-D weights / 1-D vectors, as stored) and invalidate the folded caches."""
        self.vars[full_name] = torch.as_tensor(value, dtype=torch.float32).to(self.device).contiguous()
        self._folded.clear()

    strict = False

    def get(self, name, shape, kind):
        full = self.path(name)
        if full in self.vars and tuple(self.vars[full].shape) != tuple(shape):
            raise ValueError(f"variable {full}: stored shape {tuple(self.vars[full].shape)} but the layer needs {tuple(shape)}")
        if full not in self.vars and self.strict:
            raise KeyError(f"variable {full} {tuple(shape)} is not in the loaded checkpoint (VariableStore.load(strict=True))")
        if full not in self.vars:
            rng = self._rng(full)
            if kind == "xavier":  # tf.contrib.layers.xavier_initializer (uniform): limit = sqrt(6/(fan_in+fan_out))
                fan_in, fan_out = shape[0], shape[1]
                lim = math.sqrt(6.0 / (fan_in + fan_out))
                v = rng.uniform(-lim, lim, size=shape)
            elif kind == "zeros":
                v = np.zeros(shape)
            elif kind == "ones":
                v = np.ones(shape)
            elif kind == "small":  # randomised BN statistics / biases for tests
                v = rng.uniform(-0.2, 0.2, size=shape)
            elif kind == "positive":
                v = rng.uniform(0.5, 1.5, size=shape)
            else:
                raise ValueError(kind)
            self.vars[full] = torch.tensor(v, dtype=torch.float32, device=self.device)
        return self.vars[full]

    def layer(self, cin, cout, bn, weight_decay=None):
        """(W', b') of the layer at the current scope, BN folded, cached.  weight_decay: as the reference's
        _variable_with_weight_decay, a layer built with one contributes wd * l2_loss(weights) to get_loss()."""
        if weight_decay is not None:
            self.decays[self.path("weights")] = float(weight_decay)
        key = self.path("")
        if key not in self._folded:
            w = self.get("weights", (cin, cout), "xavier")
            b = self.get("biases", (cout,), "small" if self.randomize_bn else "zeros")
            if bn:
                with self.scope("bn"):
                    gamma = self.get("gamma", (cout,), "positive" if self.randomize_bn else "ones")
                    beta = self.get("beta", (cout,), "small" if self.randomize_bn else "zeros")
                    mean = self.get("moving_mean", (cout,), "small" if self.randomize_bn else "zeros")
                    var = self.get("moving_variance", (cout,), "positive" if self.randomize_bn else "ones")
                s = gamma / torch.sqrt(var + BN_EPS)
                w, b = w * s, (b - mean) * s + beta
            self._folded[key] = (w.contiguous(), b.contiguous())
        return self._folded[key]

    def export_numpy(self):
        """scope -> {"w","b"[, "gamma","beta","mean","var"]} as the oracle expects (oracle/cells.py)."""
        out = {}
        for full, t in self.vars.items():
            parts = full.split("/")
            a = t.detach().cpu().numpy()
            if parts[-2:-1] == ["bn"]:
                sc = "/".join(parts[:-2])
                key = {"gamma": "gamma", "beta": "beta", "moving_mean": "mean", "moving_variance": "var"}[parts[-1]]
            else:
                sc = "/".join(parts[:-1])
                key = {"weights": "w", "biases": "b"}[parts[-1]]
            out.setdefault(sc, {})[key] = a
        return out


_STORE = None


def set_store(store):
    global _STORE
    _STORE = store
    return store


def store():
    if _STORE is None:
        raise RuntimeError("no VariableStore active: call tf_util.set_store(VariableStore(seed)) first")
    return _STORE


def variable_scope(name):
    return store().scope(name)


def _require_inference(is_training):
    if is_training not in (False, None):
        raise NotImplementedError("pointasnl_amd mirrors the inference graph only (is_training must be False)")


def _act(x, activation_fn):
    if activation_fn is None:
        return x
    if activation_fn in ("relu", torch.relu, torch.nn.functional.relu):
        return torch.relu_(x)
    if activation_fn in ("sigmoid", torch.sigmoid):
        return torch.sigmoid_(x)
    if activation_fn == "leaky_relu":  # tf.nn.leaky_relu default alpha = 0.2
        return torch.nn.functional.leaky_relu(x, 0.2, inplace=True)
    return activation_fn(x)


DENSE_ROWS = True     # dense layers with <= DENSE_ROWS_MAX rows (the classifier head) on pasnl_dense_rows instead of a vendor GEMM
DENSE_ROWS_MAX = 128
_DENSE_WS = {}


def _dense_rows_workspace(nbytes, device):
    """Zero-filled scratch of pasnl_dense_rows, one per (device, stream): a workspace serves one stream at a time.  Buffers
    are only ever ADDED: a captured graph keeps writing its partial sums and counters into the buffer it was captured with,
    so a buffer that proved too small for a later, larger layer stays alive (in the list) next to its replacement."""
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    bufs = _DENSE_WS.setdefault(key, [])
    if not bufs or bufs[-1].numel() < nbytes:
        bufs.append(torch.zeros(max(nbytes, 1 << 20), dtype=torch.uint8, device=device))
    return bufs[-1]


# Thin products with a long contraction on pasnl_dense_splitk -- only where it was measured to beat the vendor library WITH its
# transposed-weights trick (tools/splitk_probe.py): one column block of tiles, thousands of rows, K >= 4096, e.g. sem_seg_res's
# (2560, 4096, 128): 62 -> 44 us (at most ONE workgroup per CU: 13 slices x 20 tiles had been 260).  Everywhere else the vendor kernels are as fast or faster (EXPERIMENTS.md).
DENSE_SPLITK = True
SPLITK_MIN_K = 4096
SPLITK_MAX_N = 128
SPLITK_MIN_ROWS = 2048
SPLITK_MAX_TILES = 96
_SPLITK_WS = {}


def _dense_splitk(x2d, w, b, relu):
    """act(x2d . w + b) for a thin product with a long contraction: csrc/dense.hip (128 x 128 tiles x K slices, partial tiles
    summed in slice order by a second kernel).  The workspace is per (device, stream) and only ever grows by ADDING buffers
    (a captured graph keeps the one it was captured with)."""
    import ctypes
    rows, cin = x2d.shape
    cout = w.shape[1]
    if x2d.stride(1) != 1:
        x2d = x2d.contiguous()
    nbytes = int(_hip.lib().pasnl_dense_splitk_workspace_bytes(rows, cin, cout))
    key = (x2d.device.index, torch.cuda.current_stream().cuda_stream)
    bufs = _SPLITK_WS.setdefault(key, [])
    if not bufs or bufs[-1].numel() < nbytes:
        bufs.append(torch.empty(max(nbytes, 1 << 22), dtype=torch.uint8, device=x2d.device))
    ws = bufs[-1]
    out = torch.empty((rows, cout), dtype=torch.float32, device=x2d.device)
    _hip.launch("pasnl_dense_splitk", "dense_splitk", rows, cin, cout, int(x2d.stride(0)), _hip.ptr(x2d), _hip.ptr(w), _hip.ptr(b),
                int(bool(relu)), _hip.ptr(out), _hip.ptr(ws), ctypes.c_size_t(ws.numel()))
    return out


# fp32-grade products on the bf16 matrix pipe (csrc/dense_bf16x3.hip): an explicit MODE, off by default and off for the
# headline benchmark -- every fp32 operand as three bf16 terms, six products, fp32 accumulation: ~2x the fp32 chain's rounding
# error (inside the 1e-5 contract), not the same bits.  Taken by the long GEMMs only (after_conv / decode_after_conv windows).
DENSE_BF16X3 = False
BF16X3_MIN_TILES = 256  # 128 x 128 output tiles: one per CU at least (below that the vendor's split-K kernels win: x0.4 .. 0.9)
BF16X3_MIN_K = 512


def _dense_bf16x3(x2d, w, b, relu):
    """act(x2d . w + b) on pasnl_dense_bf16x3; the split weights are cached in the store next to the folded ones."""
    rows, cin = x2d.shape
    cout = w.shape[1]
    st = store()
    key = "@bf16x3:%x" % w.data_ptr()
    if key not in st._folded:
        nbytes = int(_hip.lib().pasnl_bf16x3_weights_bytes(cin, cout))
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=w.device)
        wc = w.contiguous()
        _hip.launch("pasnl_bf16x3_split_weights", "bf16x3_split", cin, cout, _hip.ptr(wc), _hip.ptr(ws))
        st._folded[key] = (ws, w)  # (keeps `w` alive: the pointer in the key stays unique)
    ws = st._folded[key][0]
    out = torch.empty((rows, cout), dtype=torch.float32, device=x2d.device)
    _hip.launch("pasnl_dense_bf16x3", "dense_bf16x3", rows, cin, cout, int(x2d.stride(0)), _hip.ptr(x2d), _hip.ptr(ws), _hip.ptr(b),
                int(bool(relu)), _hip.ptr(out))
    return out


def _dense_rows(x2d, w, b, relu):
    """act(x2d . w + b) for a handful of rows: csrc/dense.hip (K slices over ~128 workgroups, fixed summation order)."""
    import ctypes
    rows, cin = x2d.shape
    cout = w.shape[1]
    x2d = x2d.contiguous()
    nbytes = int(_hip.lib().pasnl_dense_rows_workspace_bytes(rows, cin, cout))
    ws = _dense_rows_workspace(nbytes, x2d.device)
    out = torch.empty((rows, cout), dtype=torch.float32, device=x2d.device)
    _hip.launch("pasnl_dense_rows", "dense_rows", rows, cin, cout, _hip.ptr(x2d), _hip.ptr(w), _hip.ptr(b), int(bool(relu)),
                _hip.ptr(out), _hip.ptr(ws), ctypes.c_size_t(ws.numel()))
    return out


def _dense(inputs, num_output_channels, scope, bn, activation_fn, weight_decay=None, input_pad=0, row_order=None):
    cin = inputs.shape[-1] - input_pad
    with variable_scope(scope):
        w, b = store().layer(cin, num_output_channels, bn, weight_decay)
        if row_order is not None:  # (name, index): the producer wrote the cin values of a row in another fixed order
            key = store().path("") + "@" + row_order[0]
            if key not in store()._folded:
                store()._folded[key] = (w[row_order[1]].contiguous(), b)
            w, b = store()._folded[key]
        if input_pad:  # the first `input_pad` input channels are alignment padding the reference's tensor does not have
            key = store().path("") + "@pad%d" % input_pad
            if key not in store()._folded:
                store()._folded[key] = (torch.cat([w.new_zeros((input_pad, w.shape[1])), w], dim=0).contiguous(), b)
            w, b = store()._folded[key]
            cin += input_pad
    x2d = inputs.reshape(-1, cin)
    is_relu = activation_fn in ("relu", torch.relu, torch.nn.functional.relu)
    if (DENSE_ROWS and x2d.is_cuda and 0 < x2d.shape[0] <= DENSE_ROWS_MAX and cin % 8 == 0 and (is_relu or activation_fn is None)
            and x2d.dtype == torch.float32):
        try:
            out = _dense_rows(x2d, w, b, is_relu)
            return out.reshape(*inputs.shape[:-1], num_output_channels)
        except _hip.PasnlUnsupported:
            pass  # e.g. an unaligned view: the vendor GEMM below
    if (DENSE_SPLITK and x2d.is_cuda and x2d.dtype == torch.float32 and (is_relu or activation_fn is None)
            and cin >= SPLITK_MIN_K and cin % 16 == 0 and num_output_channels % 4 == 0
            and num_output_channels <= SPLITK_MAX_N and x2d.shape[0] >= SPLITK_MIN_ROWS
            and -(-x2d.shape[0] // 128) * -(-num_output_channels // 128) <= SPLITK_MAX_TILES):
        try:
            out = _dense_splitk(x2d, w, b, is_relu)
            return out.reshape(*inputs.shape[:-1], num_output_channels)
        except _hip.PasnlUnsupported:
            pass  # e.g. an unaligned view: the vendor GEMM below
    if (DENSE_BF16X3 and x2d.is_cuda and x2d.dtype == torch.float32 and (is_relu or activation_fn is None)
            and cin >= BF16X3_MIN_K and cin % 32 == 0 and num_output_channels % 128 == 0
            and -(-x2d.shape[0] // 128) * (num_output_channels // 128) >= BF16X3_MIN_TILES
            and x2d.stride(1) == 1 and x2d.stride(0) % 4 == 0 and x2d.data_ptr() % 16 == 0):
        try:
            out = _dense_bf16x3(x2d, w, b, is_relu)
            return out.reshape(*inputs.shape[:-1], num_output_channels)
        except _hip.PasnlUnsupported:
            pass
    if w.shape[0] >= WEIGHTS_TRANSPOSED_MIN_K and x2d.is_cuda:
        # long contractions (the [1,C] after_conv / decode_after_conv windows: K = 2048 ... 16480): the vendor library picks
        # a better kernel when the weights are stored (N,K) and handed over as a transposed view (tools/gemm_layout_probe.py:
        # (320,16384,512) 98 -> 68 us, (8192,4096,256) 138 -> 131 us; short contractions do not care)
        key = "@T%x" % w.data_ptr()
        st = store()
        if key not in st._folded:
            st._folded[key] = (w.t().contiguous(), w)  # (keeps `w` alive: the pointer in the key stays unique)
        w = st._folded[key][0].t()
    if activation_fn in ("relu", torch.relu, torch.nn.functional.relu) and FUSE_RELU_EPILOGUE:
        # bias + ReLU in the GEMM epilogue (hipBLASLt) instead of a second pass over the output
        out = torch._addmm_activation(b, x2d, w)
        return out.reshape(*inputs.shape[:-1], num_output_channels)
    out = torch.addmm(b, x2d, w).reshape(*inputs.shape[:-1], num_output_channels)
    return _act(out, activation_fn)


def conv1d(inputs, num_output_channels, kernel_size, scope, stride=1, padding='SAME', data_format='NHWC',
           use_xavier=True, stddev=1e-3, weight_decay=None, activation_fn="relu", bn=False, bn_decay=None,
           is_training=None):
    """tf_util.py:52-117, kernel_size 1 only (all the models use)."""
    _require_inference(is_training)
    if kernel_size != 1 or data_format != 'NHWC':
        raise NotImplementedError("conv1d mirror supports kernel_size=1, NHWC")
    return _dense(inputs, num_output_channels, scope, bn, activation_fn, weight_decay)


def conv2d(inputs, num_output_channels, kernel_size, scope, stride=[1, 1], padding='SAME', data_format='NHWC',
           use_xavier=True, stddev=1e-3, weight_decay=None, activation_fn="relu", bn=False, bn_decay=None,
           is_training=None, input_pad=0, row_order=None):
    """tf_util.py:120-185.  [1,1] kernels, and the [1,W] VALID kernel that collapses the whole W axis
    (pointasnl_util.py:275, 337): inputs (B,H,W,C) -> (B,H,1,cout) == one GEMM over the flattened (W,C) window."""
    _require_inference(is_training)
    kh, kw = kernel_size
    if data_format != 'NHWC' or kh != 1:
        raise NotImplementedError("conv2d mirror supports NHWC, kernel height 1")
    if kw == 1:
        return _dense(inputs, num_output_channels, scope, bn, activation_fn, weight_decay, input_pad=input_pad)
    if padding != 'VALID' or kw != inputs.shape[2]:
        raise NotImplementedError("conv2d mirror supports [1,1] or the full-width VALID kernel")
    b, h, w, c = inputs.shape
    out = _dense(inputs.reshape(b, h, w * c), num_output_channels, scope, bn, activation_fn, weight_decay, row_order=row_order)
    return out.reshape(b, h, 1, num_output_channels)


def fully_connected(inputs, num_outputs, scope, use_xavier=True, stddev=1e-3, weight_decay=None,
                    activation_fn="relu", bn=False, bn_decay=None, is_training=None):
    """tf_util.py:327-365"""
    _require_inference(is_training)
    return _dense(inputs, num_outputs, scope, bn, activation_fn, weight_decay)


def dropout(inputs, is_training, scope, keep_prob=0.5, noise_shape=None):
    """tf_util.py:594-615: identity at inference."""
    _require_inference(is_training)
    return inputs


def l2_loss(t):
    """tf.nn.l2_loss: sum(t ** 2) / 2"""
    return (t * t).sum() * 0.5


def regularization_loss(weights_decay):
    """weights_decay * sum of l2_loss over every variable whose name contains 'weights' (the models' get_loss)."""
    st = store()
    return weights_decay * sum(l2_loss(v) for name, v in st.vars.items() if 'weights' in name)


def collection_losses(extra=()):
    """tf.add_n(tf.get_collection('losses')): the wd * l2_loss(weights) terms of the layers built with a weight_decay
    (tf_util.py:46-48) plus whatever else the graph added to that collection -- `extra`: tf.losses.* functions add their
    result to tf.GraphKeys.LOSSES, which IS the string 'losses'.  Like tf.add_n it refuses an empty collection."""
    st = store()
    terms = [wd * l2_loss(st.vars[name]) for name, wd in st.decays.items()] + list(extra)
    if not terms:
        raise ValueError("the 'losses' collection is empty: build the model with a weight_decay")
    return sum(terms)
