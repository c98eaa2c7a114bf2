"""Host-side mirrors of the reference's streaming model objects.

  AutoEncoderStreamGenerator  <->  models/autoencoder/AudioDec.py:166-256  StreamGenerator
  HiFiGANStreamGenerator      <->  models/vocoder/HiFiGAN.py:222-305       StreamGenerator

Same constructor keywords (the ``generator_params`` block of config.yml), same method names and
argument meaning (``encode / quantize / lookup / decode / initial_encoder / initial_decoder /
reset_buffer / load_state_dict / eval / to``), so ``utils/audiodec.py``-style callers work
unchanged.  All arithmetic runs in libaudiodec_hip.so; these classes only own tensors and call it.

Extension over the reference (which is batch-1 only, layers/conv_layer.py:144-146,
layers/vq_module.py:148-161): ``configure(num_streams=B, max_frames=F)`` makes one object carry B
independent streams; ``encode`` then takes ``(B, C, L)``, ``quantize`` returns ``(n_q, B, T)``,
``lookup`` returns ``(B, T, 64)``, ``decode`` returns ``(B, out, T*hop)``.  With B = 1 all shapes are
the reference's.
"""
import ctypes as C
import math
import os

import torch

from . import arch, lazy_guard, native, program

# keyword defaults of the reference constructors (AudioDec.py:169-189, HiFiGAN.py:225-241)
_AE_DEFAULTS = dict(
    input_channels=1, output_channels=1, encode_channels=32, decode_channels=32, code_dim=64,
    codebook_num=8, codebook_size=1024, bias=True, enc_ratios=(2, 4, 8, 16), dec_ratios=(16, 8, 4, 2),
    enc_strides=(3, 4, 5, 5), dec_strides=(5, 5, 4, 3), mode="causal", codec="audiodec",
    projector="conv1d", quantier="residual_vq", nonlinear_activation="ELU",
    nonlinear_activation_params={}, use_weight_norm=False)
_HG_DEFAULTS = dict(
    in_channels=80, out_channels=1, channels=512, kernel_size=7, upsample_scales=(8, 8, 2, 2),
    upsample_kernel_sizes=(16, 16, 4, 4), resblock_kernel_sizes=(3, 7, 11),
    resblock_dilations=[(1, 3, 5), (1, 3, 5), (1, 3, 5)], groups=1, bias=True, use_additional_convs=True,
    nonlinear_activation="LeakyReLU", nonlinear_activation_params={"negative_slope": 0.1},
    use_weight_norm=True, stats=None)


def _merge(defaults, kwargs, what):
    unknown = set(kwargs) - set(defaults)
    if unknown:
        raise TypeError(f"{what}.__init__() got unexpected keyword argument(s) {sorted(unknown)}")
    p = dict(defaults)
    p.update(kwargs)
    return p


class _StreamBase:
    def __init__(self):
        self._sd = None
        self._device = None
        self.num_streams = 1
        self.max_frames = 16
        self.offline = False
        self.workgroups = 0
        # arithmetic of the matrix-core convs: split-f16 operands (f16 hi + f16 lo/2048, three f16 MFMAs per product sum, f32
        # accumulation: 2^-22 relative per operand, measured error below the f32 MFMA chain's) unless ADK_SPLIT16=0 asks for the
        # exact-f32 kernels.  An operand beyond the f16 range (|v| > 65504) is caught on the device; with `guard` the step is
        # repeated on the exact-f32 twin of the program, which then stays in charge (set_guard)
        self.split16 = os.environ.get("ADK_SPLIT16", "1") == "1"
        self.guard = os.environ.get("ADK_GUARD", "1") == "1"
        self.graph = os.environ.get("ADK_GRAPH", "0") == "1"
        # steps a program can be taken back by rewind() beyond the last one: extra ring rows (HipProgram(rewind_depth=...)).  What the
        # deferred guard of pipeline.StreamingPipeline needs (it finds a failed step one to `depth` steps late); 0 = none
        self.rewind_depth = int(os.environ.get("ADK_REWIND_DEPTH", "4"))
        self._defer = None              # a list while a pipeline / the call log owns the guard: _step appends (program, frames, ticket) and does not wait
        # how a direct call is guarded: "lazy" (default) -- the check of a call is posted behind it and read when its result is first looked at
        # (lazy_guard.py: results come back as GuardedTensor; needs rewind_depth >= 1); "sync" -- one stream synchronisation per program step
        # before the call returns (rounds 3-5; ADK_GUARD_MODE=sync)
        self.guard_mode = os.environ.get("ADK_GUARD_MODE", "lazy")
        self._log = None                # lazy_guard.CallLog, made on first use or shared by the facade (share_log)
        self._replay = False            # the next steps repeat rewound ones: programs skip their ring writes (HipProgram.step(replay=True))
        self._warm = {}

    # ---- torch.nn.Module surface the reference's loader touches (bin/stream.py:59-61) ----
    def eval(self):
        return self

    def to(self, device):
        self._device = torch.device(device)
        if self._device.type == "cuda" and self._device.index is None:
            self._device = torch.device("cuda", torch.cuda.current_device() if torch.cuda.is_available() else 0)
        return self

    def configure(self, num_streams=1, max_frames=16):
        """Number of independent streams carried by this object and the largest chunk (in hops)
        one kernel sequence handles; longer calls are split, which is exact (chunked streaming ==
        one-shot)."""
        if num_streams < 1 or max_frames < 1:
            raise ValueError("num_streams and max_frames must be >= 1")
        if (num_streams, max_frames) != (self.num_streams, self.max_frames):
            self.settle()
            self.num_streams, self.max_frames = int(num_streams), int(max_frames)
            self._drop_programs()
        return self

    def set_rewind_depth(self, depth):
        """Extra ring rows for `depth` more rewindable steps (see __init__).  Rebuilds the programs: call it before the warm-up."""
        if int(depth) != self.rewind_depth:
            self.settle()
            self.rewind_depth = max(0, int(depth))
            self._drop_programs()
        return self

    def set_guard(self, on=True, mode=None):
        """(on=None: keep the current setting -- the constructor's, i.e. ADK_GUARD, default on.)
        guard=True (default): every program step is followed by a check of the program's device flag word.  mode "lazy" (default): the
        check is posted behind the step and read when the call's result is first looked at, or by a later call (lazy_guard.py) -- direct
        calls then cost what unguarded ones do; mode "sync": one stream synchronisation per step before the call returns.
        A split-f16 step that met an operand beyond the f16 range is REPEATED on the exact-f32
        kernels -- ring cursors rewound, state carried over, same inputs: exact, because a step only reads history rows earlier
        steps wrote -- and the program stays on them from then on (a warning says so); any other device-side failure raises
        here, at the step that caused it.  guard=False: nothing synchronises; failures surface at the caller's next
        native.raise_on_device_flags() (asynchronous multi-stream pipelines: bench.py).  Streaming and offline programs
        (set_offline) are repaired alike: the rewind also restores the "first step after reset" bit the offline lowering's
        replication pad depends on."""
        self.settle()
        if on is not None:
            self.guard = bool(on)
        if mode is not None:
            if mode not in ("lazy", "sync"):
                raise ValueError("guard mode must be 'lazy' or 'sync'")
            self.guard_mode = mode
        return self

    # ---- the lazy guard of direct calls (lazy_guard.py) ----
    def share_log(self, log):
        """Use `log` (a lazy_guard.CallLog) for this generator's direct calls: generators whose results feed each other on one device share
        one, so that a repair can repeat the calls that consumed a bad result (AudioDec does this for its three generators)."""
        if log is not self._log:
            self.settle()
            self._log = log
        return self

    def settle(self):
        """Every direct call made so far is verified (and repaired if need be) when this returns."""
        if self._log is not None:
            self._log.settle()

    def _call_log(self):
        """The log a direct call is recorded in, or None: guard off / mode "sync" / a pipeline owns the guard (_defer) / the rings carry no
        rows to rewind by / offline programs / the call is a repeat made by the log itself."""
        if not self.guard or self.guard_mode != "lazy" or self._defer is not None or self.rewind_depth < 1 or self.offline:
            return None
        if self._log is None:
            self._log = lazy_guard.CallLog(self._dev())
        return None if self._log.in_redo else self._log

    def _guarded(self, impl, args, progs, frames):
        log = self._call_log()
        if log is None:
            return impl(*[lazy_guard.plain(a) for a in args])
        return log.run(self, impl, args, progs() if callable(progs) else progs, frames)

    def set_split16(self, on=True):
        """Split-precision kernels for the layers that have one (default on; ADK_SPLIT16=0 flips it): f32 operands carried
        as f16 hi + f16 lo/2048, three f16 MFMAs per product sum, f32 accumulation (csrc/conv_rl16.hip).  Off =
        exact-f32 matrix-core arithmetic everywhere."""
        if bool(on) != self.split16:
            self.settle()
            self.split16 = bool(on)
            self._drop_programs()
        return self

    def set_graph(self, on=True):
        """Replay the steady state of every program of this model as HIP graphs (adk_program_set_graph): one hipGraphLaunch per
        program and step instead of ~30 kernel launches, for steps of exactly max_frames hops.  The rings are sized so that
        their cursors cycle with a short period (more history rows than the layers need; state memory grows ~1.5x).  Results are
        bit-identical to the eager path.  Default off (env ADK_GRAPH=1 flips it)."""
        if bool(on) != self.graph:
            self.settle()
            self.graph = bool(on)
            self._drop_programs()
        return self

    def set_workgroups(self, workgroups):
        """Share of the chip the stream-K conv launches of this model assume (persistent workgroups, 0 = all):
        for callers that step several models concurrently on different HIP streams."""
        self.workgroups = int(workgroups)
        for pr in self._all_programs():
            if pr is not None:
                pr.set_workgroups(self.workgroups)
        return self

    def _all_programs(self):
        return list(self._programs().values())

    def _new_program(self, make_builder):
        """make_builder(split16) -> program.Builder.  The program is lowered in this object's arithmetic and remembers how to
        lower its exact-f32 twin (HipProgram.demote)."""
        pr = program.HipProgram(make_builder(self.split16), self.num_streams, self.max_frames, self._dev(), graph=self.graph and not self.offline,
                                twin=(lambda: make_builder(False)) if self.split16 else None, rewind_depth=self.rewind_depth)
        if self.workgroups:
            pr.set_workgroups(self.workgroups)
        return pr

    def _step(self, prog, frames, ext):
        """One program step; with `guard`, checked and -- for a split-f16 range overflow -- repeated on the exact-f32 twin."""
        if self._defer is not None:
            # a pipeline owns the guard (pipeline.StreamingPipeline): the check is posted behind the step, nobody waits here
            prog.step(frames, ext, replay=self._replay)
            self._defer.append((prog, frames, prog.post_flags()))
            return
        prog.step(frames, ext, replay=self._replay)
        if not self.guard:
            return
        fl = prog.flags()
        if fl & native.FLAG_F16_OVERFLOW and prog.split16 and prog.twin_builder is not None:
            import warnings
            prog.rewind(frames)
            prog.demote()
            prog.step(frames, ext, replay=self._replay)
            fl = (fl & ~native.FLAG_F16_OVERFLOW) | prog.flags()
            warnings.warn(f"{type(self).__name__}: an operand left the f16 range (|v| > 65504) in a split-f16 conv; the step was repeated with "
                          "the exact-f32 kernels and this program continues on them", RuntimeWarning, stacklevel=3)
        native.raise_for_flags(fl, type(self).__name__)

    def set_offline(self, offline=True):
        """offline=True lowers the NON-streaming Generator.forward used by the file-level drivers
        (codecTest.py:78-95, codecStatistic.py:92-97): every utterance starts from reset_buffer() and the
        transposed convs see the replication pad of CausalConvTranspose1d.forward (conv_layer.py:189-192)
        instead of a zero pad_buffer.  Everything else (zero left-pad of CausalConv1d.forward) already
        equals streaming from the reset state."""
        if bool(offline) != self.offline:
            self.settle()
            self.offline = bool(offline)
            self._drop_programs()
        return self

    def _expected_keys(self):
        raise NotImplementedError

    def _programs(self):
        return {}

    def reset_stream(self, b, warm=True):
        """Put stream `b` back into the warmed-up state left by initial_encoder / initial_decoder
        (warm=True) or into the all-zero state of reset_buffer() (warm=False); the other streams keep
        running.  (The reference has one stream per object and can only reset everything.)"""
        if not 0 <= b < self.num_streams:
            raise IndexError(f"stream {b} out of range 0..{self.num_streams - 1}")
        self.settle()                                          # (a repair of an unverified call would rewind across the reset)
        for name, prog in self._programs().items():
            if prog is None:
                continue
            if warm and name not in self._warm:
                # configure / set_split16 / set_offline / set_stages / load_state_dict rebuild the programs and drop the
                # captured warm-up: a silent all-zero reset here would not be the state the caller asked for
                raise native.NativeError(
                    f"reset_stream(warm=True): no warmed-up state captured for program '{name}' -- run initial_encoder / "
                    "initial_decoder after the last configure()/set_*() call, or pass warm=False for the reset_buffer() state")
            prog.restore_stream_state(b, self._warm[name] if warm else None)

    def _capture_warm(self, name, prog):
        self.settle()
        self._warm[name] = prog.capture_stream_state(0)

    def load_state_dict(self, state_dict, strict=True):
        exp = self._expected_keys()
        missing = [k for k in exp if k not in state_dict]
        unexpected = [k for k in state_dict if k not in exp]
        if strict and (missing or unexpected):
            raise RuntimeError(
                f"Error(s) in loading state_dict for {type(self).__name__}:\n"
                f"\tMissing key(s) in state_dict: {missing[:8]}{'...' if len(missing) > 8 else ''}\n"
                f"\tUnexpected key(s) in state_dict: {unexpected[:8]}{'...' if len(unexpected) > 8 else ''}")
        for k, shape in exp.items():
            if k in state_dict and tuple(state_dict[k].shape) != tuple(shape):
                raise RuntimeError(f"size mismatch for {k}: copying a param with shape {tuple(state_dict[k].shape)} "
                                   f"from checkpoint, the shape in current model is {tuple(shape)}.")
        self.settle()
        self._sd = {k: v.detach().to("cpu") for k, v in state_dict.items()}
        for k, v in self._sd.items():
            if k.endswith(".pad_buffer") and bool((v != 0).any()):
                # rings hold the raw signal; a non-zero saved pad_buffer (post-activation values of a
                # previously streamed model) cannot be adopted bit-exactly -- the warm-up overwrites it
                import warnings
                warnings.warn(f"{k}: non-zero pad_buffer in checkpoint is ignored (state is rebuilt by initial_*)")
                break
        self._drop_programs()
        return self

    def _dev(self):
        if self._device is None:
            raise native.NativeError("call .to('cuda:N') before running the model")
        return native.require_gpu(self._device)

    def _conv_keys(self, specs):
        exp = {}
        for s in specs:
            if s.wn:
                exp[s.wkey("weight_g")] = (s.wshape[0], 1, 1)
                exp[s.wkey("weight_v")] = s.wshape
            else:
                exp[s.wkey("weight")] = s.wshape
            if s.bias:
                exp[s.wkey("bias")] = (s.cout,)
            if s.kind != "conv1x1":
                exp[f"{s.name}.pad_buffer"] = (1, s.cin, s.pad)
        return exp

    def _run_chunks(self, prog, src, rows_in_per_frame, ch_in, rows_out_per_frame, ch_out, frames):
        """Drive `prog` over `frames` hops in chunks of <= max_frames.  src (B, frames*rows_in, ch_in)
        contiguous; returns (B, frames*rows_out, ch_out)."""
        B = self.num_streams
        if frames <= self.max_frames:
            out = torch.empty(B, frames * rows_out_per_frame, ch_out, dtype=torch.float32, device=src.device)
            self._step(prog, frames, (src, out))
            return out
        out = torch.empty(B, frames * rows_out_per_frame, ch_out, dtype=torch.float32, device=src.device)
        f0 = 0
        while f0 < frames:
            f = min(self.max_frames, frames - f0)
            s = src[:, f0 * rows_in_per_frame:(f0 + f) * rows_in_per_frame].contiguous()
            o = torch.empty(B, f * rows_out_per_frame, ch_out, dtype=torch.float32, device=src.device)
            self._step(prog, f, (s, o))
            out[:, f0 * rows_out_per_frame:(f0 + f) * rows_out_per_frame] = o
            f0 += f
        return out


class AutoEncoderStreamGenerator(_StreamBase):
    """AudioDec streaming generator (models/autoencoder/AudioDec.py:166-256)."""

    def __init__(self, **kwargs):
        super().__init__()
        self.params = _merge(_AE_DEFAULTS, kwargs, "StreamGenerator")
        p = self.params
        if p["mode"] != "causal":      # check_mode (models/utils.py:13-15), asserted once here
            raise AssertionError(f"Mode {p['mode']} does not support AudioDec Streamer!")
        if p["codec"] not in ("audiodec", "activate_audiodec"):
            raise NotImplementedError(f"Codec ({p['codec']}) is not supported!")
        if p["projector"] != "conv1d":
            raise NotImplementedError(f"Model ({p['projector']}) is not supported!")
        if p["quantier"] != "residual_vq":
            raise NotImplementedError(f"Model ({p['quantier']}) is not supported!")
        if p["input_channels"] < 1 or p["output_channels"] < 1:
            raise ValueError("input_channels / output_channels must be positive")
        self.input_channels, self.output_channels = p["input_channels"], p["output_channels"]
        self.hop = arch.hop_length(p)
        self.n_q, self.dim, self.size = p["codebook_num"], p["code_dim"], p["codebook_size"]
        self._enc = self._dec = None
        self._embed = self._enorm = self._codebook = None

    def _drop_programs(self):
        self._enc = self._dec = None
        self._embed = self._enorm = self._codebook = None
        self._warm = {}

    def _expected_keys(self):
        p = self.params
        exp = self._conv_keys(arch.autoencoder_encoder_convs(p) + arch.autoencoder_decoder_convs(p))
        for i in range(self.n_q):
            pre = f"quantizer.codebook.layers.{i}"
            exp[f"{pre}.embed"] = (self.dim, self.size)
            exp[f"{pre}.cluster_size"] = (self.size,)
            exp[f"{pre}.embed_avg"] = (self.dim, self.size)
        return exp

    # ---- lazily built device state ----
    def _encoder(self):
        if self._enc is None:
            self._enc = self._new_program(lambda s16: program.build_encoder(self._sd, self.params, s16))
        return self._enc

    def _decoder(self):
        if self._dec is None:
            self._dec = self._new_program(lambda s16: program.build_sym_decoder(self._sd, self.params, self.offline, s16))
        return self._dec

    def _quantizer(self):
        if self._embed is None:
            dev = self._dev()
            embeds = [self._sd[f"quantizer.codebook.layers.{i}.embed"].float() for i in range(self.n_q)]
            # |E|^2 exactly as the reference forms it on the host (layers/vq_module.py:96)
            enorm = torch.stack([e.pow(2).sum(0, keepdim=True)[0] for e in embeds])
            self._embed = torch.stack(embeds).contiguous().to(dev)
            self._enorm = enorm.contiguous().to(dev)
        return self._embed, self._enorm

    def initial(self):
        """Quantizer.initial -> ResidualVQ.initial (layers/vq_module.py:151-157)."""
        dev = self._dev()
        cb = torch.stack([self._sd[f"quantizer.codebook.layers.{i}.embed"].float().transpose(0, 1) for i in range(self.n_q)])
        self._codebook = cb.reshape(-1, cb.size(-1)).contiguous().to(dev)

    # ---- reference API ----
    def initial_encoder(self, receptive_length, device):
        """AudioDec.py:216-221: warm every encoder-side state with `receptive_length` samples of silence."""
        self.to(device)
        self.initial()
        frames = math.ceil(receptive_length / self.hop)
        z = self.encode(torch.zeros(self.num_streams, self.input_channels, frames * self.hop, device=self._dev()))
        self._capture_warm("enc", self._encoder())
        idx = self.quantize(z[:1])
        return self.lookup(idx)

    def initial_decoder(self, zq):
        self.decode(zq)                                        # AudioDec.py:224-225
        self._capture_warm("dec", self._decoder())

    def _programs(self):
        return {"enc": self._enc, "dec": self._dec}

    def encode(self, x):
        """(B, C, L) -> z (B', code_dim, ceil(L/hop))   (AudioDec.py:228-234)."""
        frames = -(-int(lazy_guard.plain(x).shape[-1]) // self.hop)
        return self._guarded(self._encode, (x,), lambda: [self._encoder()], frames)

    def _encode(self, x):
        dev = self._dev()
        (batch, channel, length) = x.size()
        if channel != self.input_channels:
            x = x.reshape(-1, self.input_channels, length)
        if x.shape[0] != self.num_streams:
            raise ValueError(f"encode: got {x.shape[0]} streams, this object carries {self.num_streams} "
                             "(configure(num_streams=...))")
        x = x.to(device=dev, dtype=torch.float32)
        frames = -(-length // self.hop)
        if frames == 0:
            return torch.empty(x.shape[0], self.dim, 0, device=dev)
        if frames * self.hop != length:
            # one-shot call on a ragged length (demoFile.py:58): zero-pad to a hop multiple; every output
            # frame is exact by causality, only the state left behind differs
            x = torch.nn.functional.pad(x, (0, frames * self.hop - length))
        if self.input_channels == 1:
            src = x.reshape(x.shape[0], frames * self.hop, 1) if x.is_contiguous() else x.contiguous().reshape(x.shape[0], -1, 1)
        else:
            src = x.transpose(1, 2).contiguous()               # channel-last rows (B, L, C): what the first conv's ring holds
        z = self._run_chunks(self._encoder(), src, self.hop, self.input_channels, 1, self.dim, frames)
        return z.transpose(1, 2)

    def quantize(self, z):
        """z (B, code_dim, T) -> idx (n_q, T) for B == 1, (n_q, B, T) otherwise  (AudioDec.py:237-239)."""
        return self._guarded(self._quantize, (z,), [], 0)

    def _quantize(self, z):
        dev = self._dev()
        embed, enorm = self._quantizer()
        B, D, T = z.shape
        zt = z.to(device=dev, dtype=torch.float32).transpose(2, 1).contiguous()
        idx = torch.empty(self.n_q, B * T, dtype=torch.int64, device=dev)
        native.check(native.lib().adk_rvq_encode(
            C.c_void_p(zt.data_ptr()), C.c_void_p(embed.data_ptr()), C.c_void_p(enorm.data_ptr()),
            C.c_void_p(idx.data_ptr()), None, B * T, self.n_q, self.dim, self.size, native.current_stream(dev)),
            "adk_rvq_encode")
        idx = idx.reshape(self.n_q, B, T)
        return idx.squeeze(1) if B == 1 else idx

    def quantizer_forward(self, z):
        """Quantizer.forward in eval mode (quantizer.py:32-35 -> ResidualVQ.forward, vq_module.py:119-134):
        z (B, code_dim, T) -> zq (B, code_dim, T), the sum of the straight-through stage outputs."""
        return self._guarded(self._quantizer_forward, (z,), [], 0)

    def _quantizer_forward(self, z):
        dev = self._dev()
        embed, enorm = self._quantizer()
        B, D, T = z.shape
        zt = z.to(device=dev, dtype=torch.float32).transpose(2, 1).contiguous()
        idx = torch.empty(self.n_q, B * T, dtype=torch.int64, device=dev)
        zq = torch.empty(B, T, D, dtype=torch.float32, device=dev)
        native.check(native.lib().adk_rvq_encode(
            C.c_void_p(zt.data_ptr()), C.c_void_p(embed.data_ptr()), C.c_void_p(enorm.data_ptr()),
            C.c_void_p(idx.data_ptr()), C.c_void_p(zq.data_ptr()), B * T, self.n_q, self.dim, self.size,
            native.current_stream(dev)), "adk_rvq_encode")
        return zq.transpose(2, 1)

    def lookup(self, idx):
        """idx (n_q, T) -> zq (1, T, code_dim); (n_q, B, T) -> (B, T, code_dim)  (AudioDec.py:242-243)."""
        return self._guarded(self._lookup, (idx,), [], 0)

    def _lookup(self, idx):
        dev = self._dev()
        if self._codebook is None:
            self.initial()
        idx = idx.to(device=dev, dtype=torch.int64)
        if idx.dim() == 2:
            idx = idx.unsqueeze(1)
        n_q, B, T = idx.shape
        idx = idx.contiguous()
        zq = torch.empty(B, T, self.dim, dtype=torch.float32, device=dev)
        native.check(native.lib().adk_rvq_lookup(
            C.c_void_p(idx.data_ptr()), C.c_void_p(self._codebook.data_ptr()), C.c_void_p(zq.data_ptr()),
            B * T, n_q, self.dim, self._codebook.shape[0], native.current_stream(dev)), "adk_rvq_lookup")
        return zq

    # ---- bit-packed transport (audiodec_amd/wire.py; the reference has no wire format) ----
    def pack(self, idx, check=True):
        """Emitted indices -> uint8 payload (B, T, n_q*bits/8): 10 bytes per frame for 8 x 1024 codes."""
        from . import wire
        return wire.pack_codes(idx.to(self._dev()), self.size, check)        # (a guarded idx settles its log here: the payload is about to leave)

    def unpack(self, payload):
        from . import wire
        return wire.unpack_codes(payload.to(self._dev()), self.n_q, self.size)

    def lookup_packed(self, payload):
        """payload -> zq (B, T, code_dim): unpack fused into the codebook lookup."""
        from . import wire
        if self._codebook is None:
            self.initial()
        return wire.lookup_packed(payload.to(self._dev()), self._codebook, self.n_q, self.size)

    def decode(self, zq):
        """zq (B, T, code_dim) -> y (B, out_channels, T*hop)  (AudioDec.py:246-247)."""
        return self._guarded(self._decode, (zq,), lambda: [self._decoder()], _frames_of(zq, 1))

    def _decode(self, zq):
        return _decode_common(self, self._decoder(), zq, self.dim, self.hop, self.output_channels)

    def reset_buffer(self):
        """Zero every state ring (AudioDec.py:250-256)."""
        self.settle()
        for pr in (self._enc, self._dec):
            if pr is not None:
                pr.reset()


def _frames_of(t, rows_per_frame):
    """Hops a decode call of (B, T * rows_per_frame, C) steps its programs by (0 for anything that is not such a tensor: the call raises)."""
    t = lazy_guard.plain(t)
    return int(t.shape[1]) // max(1, rows_per_frame) if isinstance(t, torch.Tensor) and t.dim() == 3 else 0


def _decode_common(self, prog, zq, dim, hop, out_ch=1):
    dev = self._dev()
    zq = zq.to(device=dev, dtype=torch.float32)
    if zq.dim() != 3 or zq.shape[2] != dim:
        raise ValueError(f"decode: expected (B, T, {dim}), got {tuple(zq.shape)}")
    if zq.shape[0] == 1 and self.num_streams > 1:
        zq = zq.expand(self.num_streams, -1, -1)
    if zq.shape[0] != self.num_streams:
        raise ValueError(f"decode: got {zq.shape[0]} streams, this object carries {self.num_streams}")
    T = zq.shape[1]
    if T == 0:
        return torch.empty(zq.shape[0], out_ch, 0, device=dev)
    y = self._run_chunks(prog, zq.contiguous(), 1, dim, hop, out_ch, T)
    return y.reshape(zq.shape[0], 1, T * hop) if out_ch == 1 else y.transpose(1, 2)


class HiFiGANStreamGenerator(_StreamBase):
    """HiFiGAN streaming generator (models/vocoder/HiFiGAN.py:222-305)."""

    def __init__(self, **kwargs):
        super().__init__()
        self.params = _merge(_HG_DEFAULTS, kwargs, "StreamGenerator")
        p = self.params
        assert p["kernel_size"] % 2 == 1, "Kernel size must be odd number."            # HiFiGAN.py:61-63
        assert len(p["upsample_scales"]) == len(p["upsample_kernel_sizes"])
        assert len(p["resblock_dilations"]) == len(p["resblock_kernel_sizes"])
        self.norm = p["stats"] is not None
        self.hop = arch.hop_length(p)
        self.dim = p["in_channels"]
        env = os.environ.get("ADK_VOCODER_STAGES", "1")      # "1", "2" (cut at 2) or explicit cut points "1,2"
        self.cuts = [] if env == "1" else ([2] if env == "2" else [int(t) for t in env.split(",")])
        self.stages = len(self.cuts) + 1
        self._dec = None
        self._dec_parts = None

    def _drop_programs(self):
        self._dec = None
        self._dec_parts = None
        self._warm = {}

    def set_stages(self, cuts=(2,)):
        """Lower the vocoder as len(cuts)+1 programs cut in front of the upsample stages `cuts` (() = one program);
        decode() then runs them back to back, decode_stage(i, x) runs one -- callers that process a sequence of batches
        can put the stages on different HIP streams (software pipelining over batches, bench.py).  Same ops, same
        results."""
        cuts = [int(c) for c in cuts]
        if cuts != self.cuts:
            self.cuts, self.stages = cuts, len(cuts) + 1
            self._drop_programs()
        return self

    def _expected_keys(self):
        exp = self._conv_keys(arch.hifigan_convs(self.params))
        if self.norm:
            exp["mean"] = (self.dim,)
            exp["scale"] = (self.dim,)
        return exp

    def _decoder(self):
        """The whole vocoder as one program (stages == 1)."""
        if self.stages != 1:
            raise native.NativeError("this generator is lowered in several stages: use _decoder_stages()")
        if self._dec is None:
            self._dec = self._new_program(lambda s16: program.build_hifigan(self._sd, self.params, self.offline, s16))
        return self._dec

    def _decoder_stages(self):
        if self.stages == 1:
            return [self._decoder()]
        if self._dec_parts is None:
            self._dec_parts = [self._new_program(lambda s16, part=part: program.build_hifigan(self._sd, self.params, self.offline, s16, part, self.cuts))
                               for part in range(self.stages)]
        return self._dec_parts

    def initial_decoder(self, c):
        self.decode(c)                                         # HiFiGAN.py:264-265
        for i, pr in enumerate(self._decoder_stages()):
            self._capture_warm("dec" if i == 0 else f"dec{i}", pr)

    def _programs(self):
        if self.stages == 1:
            return {"dec": self._dec}
        parts = self._dec_parts or [None] * self.stages
        return {("dec" if i == 0 else f"dec{i}"): pr for i, pr in enumerate(parts)}

    def decode_stage(self, i, x):
        """Program i of a multi-stage lowering: 0 takes c (B, T, in_channels); the last returns (B, 1, T*hop); the
        hand-over tensors in between are (B, T*rate, channels) channel-last."""
        r_in = 1 if i == 0 else program.hifigan_stage_boundary(self.params, self.cuts[i - 1])[1]
        return self._guarded(lambda x_: self._decode_stage(i, x_), (x,), lambda: [self._decoder_stages()[i]], _frames_of(x, r_in))

    def _decode_stage(self, i, x):
        progs = self._decoder_stages()
        x = x.to(device=self._dev(), dtype=torch.float32)
        if i == 0:
            if x.dim() != 3 or x.shape[2] != self.dim:
                raise ValueError(f"decode: expected (B, T, {self.dim}), got {tuple(x.shape)}")
            if x.shape[0] == 1 and self.num_streams > 1:
                x = x.expand(self.num_streams, -1, -1)
            if x.shape[0] != self.num_streams:
                raise ValueError(f"decode: got {x.shape[0]} streams, this object carries {self.num_streams}")
            c_in, r_in = self.dim, 1
        else:
            c_in, r_in = program.hifigan_stage_boundary(self.params, self.cuts[i - 1])
        last = i == len(progs) - 1
        c_out, r_out = (1, self.hop) if last else program.hifigan_stage_boundary(self.params, self.cuts[i])
        T = x.shape[1] // r_in
        if T == 0:
            return torch.empty(x.shape[0], 1, 0, device=x.device) if last else torch.empty(x.shape[0], 0, c_out, device=x.device)
        y = self._run_chunks(progs[i], x.contiguous(), r_in, c_in, r_out, c_out, T)
        return y.reshape(x.shape[0], 1, T * self.hop) if last else y

    def decode(self, c):
        """c (B, T, in_channels) -> (B, 1, T*hop): norm, input conv, upsample stack, output conv, tanh
        (HiFiGAN.py:268-296)."""
        if self.stages == 1:
            return self._guarded(lambda c_: _decode_common(self, self._decoder(), c_, self.dim, self.hop), (c,), lambda: [self._decoder()], _frames_of(c, 1))
        for i in range(self.stages):
            c = self.decode_stage(i, c)
        return c

    def reset_buffer(self):
        self.settle()
        for pr in self._programs().values():
            if pr is not None:
                pr.reset()
