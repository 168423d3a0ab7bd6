"""The `ssl` and `text` steps of the reference's normalisation (src/normalization/normalize.py:65-180) with the two models on
the GPU: per clip 4-cnhubert/<name>.pt ([1, 768, T] float32) and 5-wav32k/<name> (int16, 32 kHz); per Chinese sentence
3-bert/<name>.pt ([1024, n_phones] float32) and the line of 2-name2text.txt.  File names and tensor layouts are the
reference's (src/utils/config/__init__.py:27-31), so train/dataset.py -- and the reference's own readers -- load them.

What stays outside (host-side text / audio plumbing of the reference, not arithmetic of this path): decoding the source file
to 32 kHz float (ffmpeg, src/utils/audio/__init__.py:13-32), the text cleaner / g2p (src/easevoice/text) and the tokenizer:
the callers hand over the decoded clip, and `input_ids` + `word2ph` + the cleaned `phones` / `norm_text`.

Resampling 32 kHz -> 16 kHz: the reference calls librosa.resample (normalize.py:154-156; librosa 0.9.2, uv.lock:1805-1806,
default res_type "kaiser_best" = resampy's windowed-sinc interpolator).  librosa / resampy are third-party packages that are
not under /root/reference: `resample_half` restates resampy's published algorithm for the ratio 1/2, where its
interpolation table is only ever read at every 256th entry and the interpolator degenerates to a symmetric 255-tap FIR
(64 zero crossings, roll-off 0.9475937167399596, Kaiser beta 14.769656459379492, gain 1/2) followed by taking every second
sample.  Parity of this one function is UNPINNED (no golden vector in the reference, package absent); it is checked against
its own definition and against the ideal half-band behaviour (tests/test_feature_extractors_cpu.py)."""
import os

import numpy as np
import torch

MAXX, ALPHA = 0.95, 0.5                  # normalize.py:61-62
_ROLLOFF, _BETA, _ZEROS, _PRECISION = 0.9475937167399596, 14.769656459379492, 64, 9


def _half_band_taps():
    """h[j], j = 0 .. 127: resampy's kaiser_best table (sinc_window(num_zeros=64, precision=9, kaiser(beta), rolloff)) at the
    entries a ratio of exactly 1/2 reads (index 256 j), times the sample ratio"""
    n = (2 ** _PRECISION) * _ZEROS
    step = 2 ** (_PRECISION - 1)
    j = np.arange(0, n + 1, step)[: 2 * _ZEROS]
    sinc = _ROLLOFF * np.sinc(_ROLLOFF * j / float(2 ** _PRECISION))
    taper = np.kaiser(2 * n + 1, _BETA)[n:][j]
    return 0.5 * sinc * taper


def resample_half(x: np.ndarray) -> np.ndarray:
    """32 kHz -> 16 kHz: y[t] = sum_{|j| <= 127} h[|j|] x[2 t + j], samples outside the clip are zeros, int(n / 2) outputs"""
    h = _half_band_taps()
    taps = np.concatenate([h[:0:-1], h]).astype(np.float64)            # j = -127 .. 127
    n_out = int(len(x) * 0.5)
    xp = np.concatenate([np.zeros(127), np.asarray(x, dtype=np.float64), np.zeros(128)])
    t = torch.from_numpy(xp)[None, None]
    y = torch.nn.functional.conv1d(t, torch.from_numpy(taps[::-1].copy())[None, None], stride=2)[0, 0]
    return y[:n_out].numpy().astype(np.float32)


def rescale_clip(audio32k: np.ndarray):
    """normalize.py:148-153 -> (the int16 clip written to 5-wav32k, the float clip that is resampled for HuBERT), or None for
    a clip the reference skips (peak above 2.2)"""
    tmp_max = np.abs(audio32k).max()
    if tmp_max > 2.2:
        return None
    a32 = (audio32k / tmp_max * (MAXX * ALPHA * 32768)) + ((1 - ALPHA) * 32768) * audio32k
    a32b = (audio32k / tmp_max * (MAXX * ALPHA * 1145.14)) + ((1 - ALPHA) * 1145.14) * audio32k
    return a32.astype("int16"), a32b.astype(np.float32)


class FeatureWriter:
    """writes the feature directory of one data set: `ssl(name, audio32k)` and `text(name, ...)` per item, `close()` writes
    2-name2text.txt.  hubert: feature_extractor.CNHubert (or any callable wav16k [n] -> [1, 768, T]); bert:
    feature_extractor.BertFeatures (or any object with phone_level_feature(input_ids, word2ph))."""

    def __init__(self, out_dir, hubert=None, bert=None):
        self.out_dir, self.hubert, self.bert = out_dir, hubert, bert
        self.bert_dir = os.path.join(out_dir, "3-bert")
        self.hubert_dir = os.path.join(out_dir, "4-cnhubert")
        self.wav_dir = os.path.join(out_dir, "5-wav32k")
        for d in (self.bert_dir, self.hubert_dir, self.wav_dir):
            os.makedirs(d, exist_ok=True)
        self._text_lines = []

    def ssl(self, name, audio32k) -> bool:
        """normalize.py:_name2go: False when the features came out non-finite (the reference then retries in fp32)"""
        from scipy.io import wavfile

        path = os.path.join(self.hubert_dir, name + ".pt")
        if os.path.exists(path):
            return True
        pair = rescale_clip(np.asarray(audio32k, dtype=np.float32))
        if pair is None:
            return True
        wav_int16, for_hubert = pair
        ssl = self.hubert(resample_half(for_hubert))
        if not bool(torch.isfinite(ssl).all()):
            return False
        wavfile.write(os.path.join(self.wav_dir, name), 32000, wav_int16)
        torch.save(ssl, path)
        return True

    def text(self, name, phones, word2ph, norm_text, language="zh", input_ids=None):
        """normalize.py:_process_text: the BERT feature only for Chinese items that do not have one yet; always the text line"""
        path = os.path.join(self.bert_dir, name + ".pt")
        if not os.path.exists(path) and language == "zh":
            feat = self.bert.phone_level_feature(input_ids, word2ph)
            if feat.shape[-1] != len(phones):
                raise ValueError("bert_feature and phones not match")          # normalize.py:123-124
            torch.save(feat, path)
        self._text_lines.append("%s\t%s\t%s\t%s" % (name, " ".join(phones), word2ph, norm_text))

    def close(self):
        with open(os.path.join(self.out_dir, "2-name2text.txt"), "w", encoding="utf8") as f:
            f.write("\n".join(self._text_lines) + "\n")
