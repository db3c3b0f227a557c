"""QuantizeNeuralNet -- the layer-by-layer driver with the reference's call surface
(reference: src/quantize_neural_net.py).  `main.py` of the reference does

    quantizer = QuantizeNeuralNet(model, name, batch_size, train_loader, mlp_bits=..., cnn_bits=..., ...)
    quantized_model = quantizer.quantize_network()                       (main.py:105-121)

and that works unchanged against this class.  Per layer it captures the input of the layer in the analog
and in the partially quantized network (quantize_neural_net.py:217-274), flattens conv kernels to rows
(:176) and hands the layer to StepAlgorithm._quantize_layer (:150, :180), which runs on the MI355X.
"""
import copy
import gc
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

import ctypes

from . import _lib
from .step_algorithm import PreparedColumns, StepAlgorithm
from .utils import InterruptException, extract_layers

LINEAR_MODULE_TYPE = nn.Linear
CONV2D_MODULE_TYPE = nn.Conv2d

RESULT_LOGGING_DIR = 'result_logging'
LAYER_LOGGING = False
# The reference runs a full gc.collect() three times per layer (quantize_neural_net.py:137, :212, :271) to drop the layer
# inputs.  Here every large tensor is released by reference counting the moment its last name goes (the aborted forward's
# InterruptException is handled without being bound, so its traceback -- and the activations its frames hold -- dies with
# the `except` clause; tests/test_gpu_driver_scale.py checks the peak memory), and a full collection costs ~40 ms of host
# time in a process that has imported torch: 3 x 54 layers = 6.5 s of a 12 s ResNet-50 run (bench.py --driver r50).
# True restores the reference's calls.
COLLECT_GARBAGE_PER_LAYER = False
# ... and what stands in for them by default: a model or hook that holds reference CYCLES (user models run through
# compat/run_main.py may) keeps captured activations alive until the cyclic collector's generational thresholds happen to
# trigger.  So after every layer the device memory in use is compared with the lowest level seen after any layer so far, and
# ONE full collection runs when it has grown by more than this many bytes (torch.cuda.memory_allocated: a counter, no
# synchronisation).  A driver without cycles returns to its low-water mark after every layer and never pays; one with cycles
# pays ~40 ms when -- and only when -- a couple of layers' worth of activations have piled up.  0 switches the check off.
GC_WHEN_DEVICE_MEMORY_GROWS_BY = 2 << 30
# Conv2d capture: gather the sampled patches on the GPU straight into the kernels' column layout
# (gpfq_gather_patches_f32) instead of materialising the full unfold and transposing it afterwards.
FUSED_CAPTURE = True


class QuantizeNeuralNet:
    '''Quantizes a network layer by layer with GPFQ.

    Public attributes kept from the reference (quantize_neural_net.py:81-114): analog_network,
    quantized_network, analog_network_layers, quantized_network_layers, mlp_boundary_idx,
    cnn_boundary_idx, mlp_alphabet_step_size, cnn_alphabet_step_size, ...
    '''

    def __init__(self,
                 network_to_quantize, network_name, batch_size, data_loader,
                 mlp_bits, cnn_bits,
                 ignore_layers,
                 mlp_alphabet_scalar, cnn_alphabet_scalar,
                 mlp_percentile, cnn_percentile,
                 reg, lamb, retain_rate, stochastic_quantization, device):
        self.network_name = network_name
        self.analog_network = network_to_quantize
        self.batch_size = batch_size
        self.data_loader_iter = iter(data_loader)

        # symmetric alphabet {-K..K}*step with K = 2^(bits-1)   (quantize_neural_net.py:87-88)
        self.mlp_bits, self.cnn_bits = mlp_bits, cnn_bits
        self.mlp_boundary_idx = 2 ** (mlp_bits - 1)
        self.cnn_boundary_idx = 2 ** (cnn_bits - 1)
        self.mlp_alphabet_scalar = mlp_alphabet_scalar
        self.mlp_alphabet_step_size = mlp_alphabet_scalar / self.mlp_boundary_idx      # :92
        self.cnn_alphabet_step_size = cnn_alphabet_scalar / self.cnn_boundary_idx      # :93
        self.mlp_percentile, self.cnn_percentile = mlp_percentile, cnn_percentile
        self.ignore_layers = ignore_layers
        self.retain_rate = retain_rate
        self.reg, self.lamb = reg, lamb
        self.device = device
        self.stochastic_quantization = stochastic_quantization

        self.quantized_network = copy.deepcopy(self.analog_network)
        self.analog_network_layers = []
        extract_layers(self.analog_network, self.analog_network_layers)
        self.quantized_network_layers = []
        extract_layers(self.quantized_network, self.quantized_network_layers)
        self.plan = None            # kernel family for every layer (GPFQ_PLAN_*; None = auto) -- extra
        # Optional: put the ANALOG columns of layer i+1 into the kernels' layout on a side stream while the loop of layer i
        # runs (the analog network never changes, so its input of the next layer does not depend on this layer's result):
        # the analog forward of layer i+1 runs on the main stream right before layer i is quantized, only the short
        # gather / transposition kernel goes to the side stream -- a long forward next to a cooperative launch would break
        # the co-residency the cooperative plan counts on.  Same batches, same numpy draws, same results.  OFF by default:
        # measured (bench.py --prefetch-analog, one GPU and the per-rank shapes of 2 / 4 / 8 GPUs alike) the loop kernels are
        # latency chains that lose more to a concurrent transposition than the overlap returns (330 -> 320 M weights/s on
        # the headline, 422 -> 362 on an 8-GPU rank's rows).  -- extra
        self.prefetch_analog = False
        self.stochastic_seed_base = 0   # layer i of this run draws Philox streams keyed by base + i -- extra
        # Optional callable(tag, layer_idx) invoked at the phase boundaries of every layer -- layer_begin, forward_begin /
        # capture_begin / capture_end (twice: analog, quantized network), prepare_begin / loop_begin / loop_end (the
        # native call), metrics_end, layer_end -- so that a profiler can record stream events there (bench.py --driver:
        # what main.py:120-125 times, split).  -- extra
        self.timing_hook = None
        self.layer_reports = []     # per-layer dicts (index, errors, step) -- extra, not in the reference
        self.layer_indices = []     # per-layer alphabet indices + step, what packed.save() writes -- extra

    def quantize_network(self):
        '''Quantize every non-ignored Linear/Conv2d layer in registration order; returns the quantized
        nn.Module (quantize_neural_net.py:117-214).  Biases are left untouched.'''
        todo = [i for i in range(len(self.quantized_network_layers)) if i not in self.ignore_layers]
        print(f'Layer indices to quantize {todo}')
        print(f'Total number of layers to quantize {len(todo)}')

        ahead = None                 # (layer index, raw batch, hook) of the layer whose analog capture is already under way
        side = None
        on_gpu = torch.cuda.is_available() and torch.device(self.device).type == 'cuda'
        low_water = None             # least device memory in use seen after a layer (GC_WHEN_DEVICE_MEMORY_GROWS_BY)
        self.garbage_collections = 0
        if self.prefetch_analog and torch.cuda.is_available() and torch.device(self.device).type == 'cuda':
            side = torch.cuda.Stream(device=self.device)
        for done, layer_idx in enumerate(todo):
            self._mark("layer_begin", layer_idx)
            if COLLECT_GARBAGE_PER_LAYER:
                gc.collect()
            if ahead is not None and ahead[0] == layer_idx:
                _, raw, save_input = ahead
            else:
                raw, save_input = self._capture_analog(layer_idx, None)
            ahead = None
            self._capture_quantized(layer_idx, raw, save_input)
            del raw
            analog_in, quantized_in = self._resolve(save_input.inputs[0]), self._resolve(save_input.inputs[1])
            if side is not None and done + 1 < len(todo):
                ahead = (todo[done + 1],) + self._capture_analog(todo[done + 1], side)
            print(f'\nQuantizing layer with index: {layer_idx}')
            print(f'Quantization progress: {done} out of {len(todo)-1}\n')

            analog_layer = self.analog_network_layers[layer_idx]
            W = analog_layer.weight.data
            if type(analog_layer) == LINEAR_MODULE_TYPE:
                groups, W_shape = 1, None
                step_size, K, pct = self.mlp_alphabet_step_size, self.mlp_boundary_idx, self.mlp_percentile
            elif type(analog_layer) == CONV2D_MODULE_TYPE:
                groups, W_shape = analog_layer.groups, W.shape
                print('shape of W:', W.shape)
                print('shape of analog_layer_input:', analog_in.shape)
                print('shape of quantized_layer_input:', quantized_in.shape)
                W = W.view(W.size(0), -1)           # one row per output channel, channel-major (:176)
                step_size, K, pct = self.cnn_alphabet_step_size, self.cnn_boundary_idx, self.cnn_percentile
            else:
                raise TypeError(f'The layer type {type(analog_layer)} is not currently supported')

            # the native counterpart of StepAlgorithm._quantize_layer (:150, :180); its result dict also carries the
            # alphabet indices and the step, so nothing travels through class state and two quantizers can interleave
            print(f'The number of groups: {groups}\n')
            res = StepAlgorithm._quantize_layer_ex(W, analog_in, quantized_in, analog_in.shape[0], step_size, K, pct,
                                                   self.reg, self.lamb, groups, self.stochastic_quantization,
                                                   self.device, plan=self.plan,
                                                   seed=self.stochastic_seed_base + done,
                                                   event_hook=(lambda tag, shape, li=layer_idx: self._mark(tag, li))
                                                   if self.timing_hook else None)
            self._mark("metrics_end", layer_idx)
            Q, quantize_error, relative_quantize_error = res["Q"], res["quantize_error"], res["relative_quantize_error"]
            quantize_adder, relative_adder = res["quantize_adder"], res["relative_adder"]
            Q = Q.float() if W_shape is None else Q.reshape(W_shape).float()
            self.quantized_network_layers[layer_idx].weight.data = Q

            print(f'The quantization error of layer {layer_idx} is {quantize_error.cpu().numpy()}.')
            print(f'The relative quantization error of layer {layer_idx} is {relative_quantize_error.cpu().numpy()}.\n')
            if res["timeouts"]:
                # a cooperative launch gave up waiting for a peer workgroup (the card is shared, or co-residency broke)
                # and the layer was redone on the plan that waits for nobody: correct, but slow -- say so
                print(f'WARNING: layer {layer_idx}: {len(res["timeouts"])} cooperative launch(es) timed out and were '
                      f'redone on the whole-row streaming plan (rows, d, m): {res["timeouts"]}')
            self.layer_reports.append(dict(layer=layer_idx, quantize_error=float(quantize_error),
                                           relative_quantize_error=float(relative_quantize_error),
                                           timeouts=list(res["timeouts"])))
            mode = 1 if self.reg == 'L1' else 2 if self.reg == 'L0' else 3 if self.stochastic_quantization else 0
            self.layer_indices.append(dict(layer=layer_idx, idx=res["idx"].detach().cpu(), step=float(res["step"]),
                                           K=int(K), mode=mode, lamb=float(self.lamb if self.lamb is not None else 0.0)))
            if LAYER_LOGGING:
                self._log_layer(layer_idx, W, Q, quantize_adder, relative_adder)

            del analog_in, quantized_in, res, Q, quantize_adder, relative_adder
            if COLLECT_GARBAGE_PER_LAYER:
                gc.collect()
            elif on_gpu and GC_WHEN_DEVICE_MEMORY_GROWS_BY > 0:
                in_use = torch.cuda.memory_allocated(self.device)
                if low_water is not None and in_use - low_water > GC_WHEN_DEVICE_MEMORY_GROWS_BY:
                    gc.collect()
                    self.garbage_collections += 1
                    in_use = torch.cuda.memory_allocated(self.device)
                low_water = in_use if low_water is None else min(low_water, in_use)
            self._mark("layer_end", layer_idx)
        return self.quantized_network

    def _mark(self, tag, layer_idx):
        if self.timing_hook is not None:
            self.timing_hook(tag, layer_idx)

    def _log_layer(self, layer_idx, W, Q, quantize_adder, relative_adder):
        '''Optional .npy dumps of W, Q, U^T and the per-neuron relative error (quantize_neural_net.py:199-209).'''
        os.makedirs(RESULT_LOGGING_DIR, exist_ok=True)
        tag = (f'batch_size:{self.batch_size}_model_name:{self.network_name}_bits:{self.mlp_bits}'
               f'_scalar:{self.mlp_alphabet_scalar}_stochastic:{self.stochastic_quantization}_layer:{layer_idx}')
        for suffix, val in (('ori', W), ('quant', Q), ('adder', quantize_adder), ('relative', relative_adder)):
            arr = val.detach().cpu().numpy() if isinstance(val, torch.Tensor) else val
            np.save(os.path.join(RESULT_LOGGING_DIR, f'{tag}_{suffix}.npy'), arr)

    def _make_hook(self, layer_idx):
        analog_layer = self.analog_network_layers[layer_idx]
        if type(analog_layer) == LINEAR_MODULE_TYPE:
            return SaveInputMLP()
        if type(analog_layer) == CONV2D_MODULE_TYPE:
            return SaveInputConv2d(kernel_size=analog_layer.kernel_size, dilation=analog_layer.dilation,
                                   padding=analog_layer.padding, stride=analog_layer.stride,
                                   groups=analog_layer.groups, retain_rate=self.retain_rate)
        raise TypeError(f'The layer type {type(analog_layer)} is not currently supported')

    def _forward_to(self, net, layer, hook, raw_input_data):
        if self.timing_hook is not None:
            inner, li = hook, self._layer_index_of(layer)

            def hook(module, module_in, module_out):
                self._mark("capture_begin", li)
                try:
                    return inner(module, module_in, module_out)
                finally:
                    self._mark("capture_end", li)
            self._mark("forward_begin", li)
        handle = layer.register_forward_hook(hook)
        try:
            with torch.no_grad():
                net(raw_input_data)
        except InterruptException:
            pass
        finally:
            handle.remove()

    def _layer_index_of(self, layer):
        for layers in (self.analog_network_layers, self.quantized_network_layers):
            for i, l in enumerate(layers):
                if l is layer:
                    return i
        return -1

    def _capture_analog(self, layer_idx, side_stream):
        '''First half of quantize_neural_net.py:217-274: the NEXT batch of the loader through the analog network, cut at
        the hooked layer.  side_stream: the hook puts its capture into the kernels' column layout on that stream (behind
        the forward, which stays on the current stream).  Returns (raw batch on the device, hook).'''
        raw_input_data, _ = next(self.data_loader_iter)
        raw_input_data = raw_input_data.to(self.device)
        save_input = self._make_hook(layer_idx)
        save_input.side_stream = side_stream
        self._forward_to(self.analog_network, self.analog_network_layers[layer_idx], save_input, raw_input_data)
        save_input.side_stream = None
        return raw_input_data, save_input

    def _capture_quantized(self, layer_idx, raw_input_data, save_input):
        '''Second half: the SAME batch through the partially quantized network with the SAME hook object (shared patch
        sample).'''
        self._forward_to(self.quantized_network, self.quantized_network_layers[layer_idx], save_input, raw_input_data)

    @staticmethod
    def _resolve(captured):
        '''A capture made on the side stream carries the event that marks it ready: the current stream waits for it.'''
        ev = getattr(captured, "ready_event", None)
        if ev is not None:
            cur = torch.cuda.current_stream(captured.device)
            cur.wait_event(ev)
            captured.T.record_stream(cur)
            captured.ready_event = None
        return captured

    def _populate_linear_layer_input(self, layer_idx):
        '''Inputs of layer `layer_idx` in the analog and in the (partially) quantized network for the NEXT
        batch of the loader (quantize_neural_net.py:217-274).  Both forwards are cut at the hooked layer.
        Returns (analog_layer_input, quantized_layer_input), each (m, features).'''
        raw_input_data, save_input = self._capture_analog(layer_idx, None)
        self._capture_quantized(layer_idx, raw_input_data, save_input)
        del raw_input_data
        if COLLECT_GARBAGE_PER_LAYER:
            gc.collect()
        return (save_input.inputs[0], save_input.inputs[1])


def _on_side_stream(side, src, make, *also):
    """Run make() -- a short kernel that reads `src` (and `also`) and returns PreparedColumns -- on the side stream, behind
    everything the current stream has queued (the forward that produced src); the result carries the event that marks it
    ready (QuantizeNeuralNet._resolve)."""
    cur = torch.cuda.current_stream(src.device)
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        out = make()
        ev = torch.cuda.Event()
        ev.record(side)
    for t in (src,) + also:
        t.record_stream(side)       # (allocated on the current stream, read on the side stream)
    out.ready_event = ev
    return out


def _pair(v):
    return (int(v[0]), int(v[1])) if isinstance(v, (tuple, list)) else (int(v), int(v))


class SaveInputMLP:
    '''Forward hook that records the input of a Linear layer and aborts the forward
    (quantize_neural_net.py:277-292).'''

    def __init__(self):
        self.inputs = []
        self.side_stream = None     # set by the driver: put this capture into the column layout on that stream

    def __call__(self, module, module_in, module_out):
        if len(module_in) != 1:
            raise TypeError('The number of input layer is not equal to one!')
        x = module_in[0]
        if self.side_stream is not None and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2:
            x = _on_side_stream(self.side_stream, x, lambda: StepAlgorithm.prepare_columns(x))
        self.inputs.append(x)
        raise InterruptException


class SaveInputConv2d:
    '''Forward hook for Conv2d layers (quantize_neural_net.py:295-350): the input feature map is cut into
    kernel-sized patches on a grid whose stride is the KERNEL SIZE (the layer's own stride is not used,
    :320), every patch becomes a row (channel-major, then kh, kw), and per image int(p*L + 1) patches
    (all L if p == 1) are drawn with replacement; the draw of the first call is reused for the second.'''

    def __init__(self, kernel_size, dilation, padding, stride, groups, retain_rate):
        self.p = retain_rate
        self.kernel_size, self.dilation, self.padding = kernel_size, dilation, padding
        self.groups = groups
        self.inputs = []
        self.call_count = 0
        self.rand_indices = None
        self.side_stream = None     # set by the driver: run the patch gather of this capture on that stream

    def __call__(self, module, module_in, module_out):
        if len(module_in) != 1:
            raise TypeError('The number of input layer is not equal to one!')
        x = module_in[0]                                               # (B, C, H, W)
        B, C, H, W = x.shape
        kh, kw = _pair(self.kernel_size)
        ph, pw = _pair(self.padding)
        dh, dw = _pair(self.dilation)
        # blocks per image of nn.Unfold(kernel, dilation, padding, stride = kernel)
        L = ((H + 2 * ph - dh * (kh - 1) - 1) // kh + 1) * ((W + 2 * pw - dw * (kw - 1) - 1) // kw + 1)
        if self.call_count == 0:
            keep = int(self.p * L + 1 if self.p != 1 else self.p * L)
            # same generator consumption as np.random.choice(np.arange(L*i, L*(i+1)), size=keep), i = 0..B-1
            self.rand_indices = np.concatenate([L * i + np.random.choice(L, size=keep) for i in range(B)])
        self.call_count += 1
        sel = torch.as_tensor(self.rand_indices, device=x.device, dtype=torch.long)
        if FUSED_CAPTURE and x.is_cuda and x.dtype == torch.float32:
            m = sel.numel()
            mp = _lib.lib.gpfq_padded_m(m)
            xc = x.contiguous()

            def gather():
                T = torch.empty((C * kh * kw, mp), device=x.device, dtype=torch.float32)
                _lib.check(_lib.lib.gpfq_gather_patches_f32(
                    ctypes.c_void_p(xc.data_ptr()), B, C, H, W, kh, kw, ph, pw, dh, dw, ctypes.c_void_p(sel.data_ptr()), m,
                    ctypes.c_void_p(T.data_ptr()), mp, _lib.current_stream_ptr(x.device)))
                return PreparedColumns(T, m)
            if self.side_stream is not None:
                self.inputs.append(_on_side_stream(self.side_stream, xc, gather, sel))
            else:
                self.inputs.append(gather())
        else:
            cols = F.unfold(x, self.kernel_size, dilation=self.dilation, padding=self.padding,
                            stride=self.kernel_size)                   # (B, C*kh*kw, L)
            rows = cols.transpose(1, 2).reshape(B * L, -1)             # (B*L, C*kh*kw)
            self.inputs.append(rows.index_select(0, sel))
        raise InterruptException
