"""File-level evaluation drivers on the HIP path (mirrors /root/reference/codecTest.py,
codecStatistic.py and bin/test.py).

The reference's drivers do not stream: they call the models' ``forward`` pieces
(``encoder`` -> ``projector`` -> ``quantizer`` -> ``decoder`` / the vocoder's ``__call__``) on a whole
utterance.  For the causal models that is exactly the streaming arithmetic started from
``reset_buffer()`` -- except the transposed convs, whose ``forward`` left-pads by replication
(layers/conv_layer.py:189-192).  ``set_offline(True)`` on the generators lowers that variant
(ADK_OP_HIST_REPLICATE), so the utterance runs through the same kernels, in chunks of ``max_frames``
hops, with every stream's state reset per utterance.

Same class names, method names, argument meaning and error behaviour as the reference drivers:
``TestMain(args).load_dataset / load_encoder / load_decoder / initial_folder / encode / decode / run``
and ``StatisticMain(args).load_dataset / load_analyzer / audio_analysis / run``.  WAV I/O goes through
``scipy.io.wavfile`` (``soundfile`` is not a dependency here).
"""
import fnmatch
import logging
import os
import sys
import time

import numpy as np
import torch
import yaml

from . import native
from .stream_generator import AutoEncoderStreamGenerator as generator_audiodec
from .stream_generator import HiFiGANStreamGenerator as generator_hifigan


# ---- dataloader/dataset.py:15-90 (SingleDataset with load_fn = sf.read(..., always_2d=True)) ----
def read_wav(path):
    """(T, C) float64 in [-1, 1), the convention of soundfile.read(always_2d=True)."""
    from scipy.io import wavfile
    _, data = wavfile.read(path)
    if data.dtype == np.int16:
        x = data.astype(np.float64) / 32768.0
    elif data.dtype == np.int32:
        x = data.astype(np.float64) / 2147483648.0
    elif data.dtype == np.uint8:
        x = (data.astype(np.float64) - 128.0) / 128.0
    else:
        x = data.astype(np.float64)
    return x[:, None] if x.ndim == 1 else x


def write_wav_pcm16(path, audio, sample_rate):
    """bin/test.py:107-113 (soundfile.write(..., "PCM_16")): scale by 0x7FFF, round to nearest, clip."""
    from scipy.io import wavfile
    pcm = np.clip(np.rint(np.asarray(audio, np.float64) * 32767.0), -32768, 32767).astype(np.int16)
    wavfile.write(path, int(sample_rate), pcm)


def find_files(root_dir, query="*.wav", include_root_dir=True):
    files = []
    for root, _, filenames in os.walk(root_dir, followlinks=True):
        for filename in fnmatch.filter(filenames, query):
            files.append(os.path.join(root, filename))
    if not include_root_dir:
        files = [f.replace(root_dir + "/", "") for f in files]
    return files


class SingleDataset:
    def __init__(self, files, query="*.wav", load_fn=read_wav, return_utt_id=False, subset_num=-1):
        self.return_utt_id, self.load_fn = return_utt_id, load_fn
        if isinstance(files, list):
            filenames = files
        elif os.path.isdir(files):
            filenames = sorted(find_files(files, query))
        elif os.path.isfile(files):
            with open(files) as f:
                filenames = sorted(line.strip() for line in f if len(line.strip()))
        else:
            raise ValueError(f"{files} is not a list / existing folder or file!")
        if subset_num > 0:
            filenames = filenames[:subset_num]
        assert len(filenames) != 0, "File list in empty!"
        self.filenames = filenames
        self.utt_ids = [os.path.splitext(os.path.basename(f))[0] for f in filenames]

    def __len__(self):
        return len(self.filenames)

    def __getitem__(self, idx):
        data = self.load_fn(self.filenames[idx])
        return (self.utt_ids[idx], data) if self.return_utt_id else data

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]


def _load_config(checkpoint, config_name="config.yml"):
    with open(os.path.join(os.path.dirname(checkpoint), config_name)) as f:       # bin/utils.py:17-22
        return yaml.load(f, Loader=yaml.Loader)


def _setup_logging():
    logging.basicConfig(level=logging.INFO, stream=sys.stdout,
                        format="%(asctime)s (%(module)s:%(lineno)d) %(levelname)s: %(message)s")


def _device():
    # the reference falls back to the CPU (bin/test.py:38-43); this path has no CPU implementation
    if not torch.cuda.is_available():
        raise native.NativeError("the offline drivers run the HIP kernels: no HIP device is visible")
    logging.info("device: gpu")
    return torch.device("cuda", torch.cuda.current_device())


def _load_generator(cls, config, checkpoint, device, max_frames):
    g = cls(**config["generator_params"])
    g.load_state_dict(torch.load(checkpoint, map_location="cpu")["model"]["generator"])
    return g.eval().to(device).configure(1, max_frames).set_offline(True)


def _streams_of(audio, multi_channel):
    if multi_channel:
        raise NotImplementedError("only mono models (input_channels = 1) are lowered")
    x = torch.tensor(audio, dtype=torch.float)
    return x.transpose(1, 0).unsqueeze(1)                    # (T, C) -> (C, 1, T)   codecTest.py:82-83


class TestMain:
    """codecTest.py:22-118 + bin/test.py:26-118."""

    def __init__(self, args, max_frames=64):
        _setup_logging()
        self.device = _device()
        self.max_frames = max_frames
        self.encoder_checkpoint = args.encoder
        self.encoder_config = _load_config(args.encoder)
        self.decoder_checkpoint = args.decoder
        self.decoder_config = _load_config(args.decoder)
        self.encoder = self.decoder = self.dataset = self.outdir = None
        self.encoder_type = self.encoder_config.get("model_type", "symAudioDec")
        self.decoder_type = self.decoder_config.get("model_type", "symAudioDec")
        self.multi_channel = self.encoder_config["generator_params"].get("input_channels", 1) > 1

    def load_dataset(self, subset, subset_num):
        data_path = os.path.join(self.encoder_config["data"]["path"], self.encoder_config["data"]["subset"][subset])
        assert os.path.exists(data_path), f"{data_path} does not exist!"
        self.dataset = SingleDataset(files=data_path, query="*.wav", return_utt_id=True, subset_num=subset_num)
        logging.info(f"The number of utterances = {len(self.dataset)}.")

    def load_encoder(self):
        if self.encoder_type not in ["symAudioDec", "symAudioDecUniv"]:
            raise NotImplementedError(f"Encoder {self.encoder_type} is not supported!")
        self.encoder = _load_generator(generator_audiodec, self.encoder_config, self.encoder_checkpoint,
                                       self.device, self.max_frames)
        logging.info(f"Loaded Encoder from {self.encoder_checkpoint}.")

    def load_decoder(self):
        if self.decoder_type in ["symAudioDec", "symAudioDecUniv"]:
            decoder = generator_audiodec
        elif self.decoder_type in ["HiFiGAN", "UnivNet"]:
            decoder = generator_hifigan
        else:
            raise NotImplementedError(f"Decoder {self.decoder_type} is not supported!")
        self.decoder = _load_generator(decoder, self.decoder_config, self.decoder_checkpoint, self.device, self.max_frames)
        logging.info(f"Loaded Decoder from {self.decoder_checkpoint}.")

    def encode(self, audio):
        """audio (T, C) -> zq (C, code_dim, T')   (codecTest.py:78-88)."""
        x = _streams_of(audio, self.multi_channel).to(self.device)
        if x.shape[0] != self.encoder.num_streams:
            self.encoder.configure(x.shape[0], self.max_frames)
        self.encoder.reset_buffer()
        z = self.encoder.encode(x)                           # encoder.encoder + encoder.projector
        return self.encoder.quantizer_forward(z)             # encoder.quantizer(z)[0]

    def decode(self, zq):
        """zq (B, code_dim, T') -> y (B, 1, T' * hop)   (codecTest.py:90-95)."""
        if zq.shape[0] != self.decoder.num_streams:
            self.decoder.configure(zq.shape[0], self.max_frames)
        self.decoder.reset_buffer()
        return self.decoder.decode(zq.transpose(2, 1))

    def initial_folder(self, subset, output_name, specific_folder="False"):
        if specific_folder == "True":
            self.outdir = output_name
        else:
            encoder = os.path.dirname(self.encoder_checkpoint).split("/")[-1]
            decoder = os.path.dirname(self.decoder_checkpoint).split("/")[-1]
            encoder_checkpoint = os.path.basename(self.encoder_checkpoint).split("steps")[0].split("-")[-1]
            decoder_checkpoint = os.path.basename(self.decoder_checkpoint).split("steps")[0].split("-")[-1]
            testdir = f"{encoder}-{decoder}_{encoder_checkpoint}-{decoder_checkpoint}"
            setdir = self.encoder_config["data"]["subset"][subset]
            self.outdir = os.path.join(output_name, testdir, setdir)
        if not os.path.exists(self.outdir):
            os.makedirs(self.outdir, exist_ok=True)

    def run(self):
        """bin/test.py:86-104: per-utterance RTF = wall time / audio duration, averaged over utterances."""
        total_rtf, idx = 0.0, 0
        with torch.no_grad():
            for idx, (utt_id, x) in enumerate(self.dataset, 1):
                start = time.time()
                zq = self.encode(x)
                y = self.decode(zq)
                y = y.squeeze(1).transpose(1, 0).cpu().numpy()                 # T x C
                native.raise_on_device_flags(f"utterance {utt_id}")           # device-side failures -> exceptions
                rtf = (time.time() - start) / (len(y) / self.decoder_config["sampling_rate"])
                total_rtf += rtf
                write_wav_pcm16(os.path.join(self.outdir, f"{utt_id}_output.wav"), y, self.decoder_config["sampling_rate"])
        self.mean_rtf = total_rtf / idx
        logging.info("Finished generation of %d utterances (RTF = %.03f)." % (idx, self.mean_rtf))
        return self.mean_rtf


def _partial_fit(state, X):
    """sklearn.preprocessing.StandardScaler.partial_fit (its _incremental_mean_and_var, float64), used when
    scikit-learn is not importable."""
    X = np.asarray(X, np.float64)
    n_new = X.shape[0]
    new_sum = X.sum(axis=0)
    if state is None:
        last_mean, last_var, last_n = 0.0, 0.0, 0
    else:
        last_mean, last_var, last_n = state
    last_sum = last_mean * last_n
    n = last_n + n_new
    mean = (last_sum + new_sum) / n
    T = new_sum / n_new
    temp = X - T
    correction = temp.sum(axis=0)
    new_unnorm = (temp ** 2).sum(axis=0) - correction ** 2 / n_new
    if last_n == 0:
        unnorm = new_unnorm
    else:
        last_unnorm = last_var * last_n
        ratio = last_n / n_new
        unnorm = last_unnorm + new_unnorm + ratio / n * (last_sum / ratio - new_sum) ** 2
    return mean, unnorm / n, n


class StatisticMain:
    """codecStatistic.py:27-113: mean / scale of the quantised code vectors over a training subset."""

    def __init__(self, args, max_frames=64):
        _setup_logging()
        self.device = _device()
        self.max_frames = max_frames
        with open(args.config, "r") as f:
            self.config = yaml.load(f, Loader=yaml.FullLoader)
        self.stats_path = self.config["stats"]
        self.analyzer_checkpoint = self.config["analyzer"]
        self.analyzer_config = _load_config(self.analyzer_checkpoint)
        self.model_type = self.analyzer_config.get("model_type", "symAudioDec")
        os.makedirs(os.path.dirname(self.stats_path), exist_ok=True)

    def load_dataset(self, subset, subset_num):
        audio_path = os.path.join(self.config["data"]["path"], self.config["data"]["subset"][subset])
        assert os.path.exists(audio_path), f"{audio_path} does not exist!"
        self.dataset = SingleDataset(files=audio_path, query="*.wav", return_utt_id=False, subset_num=subset_num)
        logging.info(f"The number of {subset} audio files = {len(self.dataset)}.")

    def load_analyzer(self):
        if self.model_type not in ["symAudioDec", "symAudioDecUniv"]:
            raise NotImplementedError(f"Analyzer {self.model_type} is not supported!")
        self.analyzer = _load_generator(generator_audiodec, self.analyzer_config, self.analyzer_checkpoint,
                                        self.device, self.max_frames)
        logging.info(f"Loaded Analyzer from {self.analyzer_checkpoint}.")

    def audio_analysis(self, audio):
        """audio (T, C) -> zq (T', code_dim) numpy   (codecStatistic.py:92-97; the reference feeds (1, C, T),
        i.e. it assumes C == input_channels == 1 here)."""
        x = torch.tensor(audio, dtype=torch.float).to(self.device)
        x = x.transpose(1, 0).unsqueeze(0)                   # (T, C) -> (1, C, T)
        if x.shape[1] != 1:
            raise NotImplementedError("only mono models (input_channels = 1) are lowered")
        if self.analyzer.num_streams != 1:
            self.analyzer.configure(1, self.max_frames)
        self.analyzer.reset_buffer()
        zq = self.analyzer.quantizer_forward(self.analyzer.encode(x))
        out = zq.squeeze(0).transpose(1, 0).cpu().numpy()    # (T', C)
        native.raise_on_device_flags("StatisticMain.audio_analysis")
        return out

    def run(self):
        try:
            from sklearn.preprocessing import StandardScaler
            scaler, state = StandardScaler(), None
        except Exception:                                    # pragma: no cover
            scaler, state = None, None
        idx = 0
        with torch.no_grad():
            for idx, x in enumerate(self.dataset, 1):
                zq = self.audio_analysis(x)
                if scaler is not None:
                    scaler.partial_fit(zq)
                else:
                    state = _partial_fit(state, zq)
        if scaler is not None:
            mean, scale = scaler.mean_, scaler.scale_
        else:
            mean, var, _ = state
            scale = np.sqrt(var)
            scale[scale < 10 * np.finfo(np.float64).eps] = 1.0        # sklearn _handle_zeros_in_scale
        stats = np.stack([mean, scale], axis=0)
        np.save(self.stats_path, stats.astype(np.float32), allow_pickle=False)
        logging.info(f"Finished statistical calculation of {idx} utterances.")
        return stats.astype(np.float32)
