#!/usr/bin/env python
"""Create a dataset of Mel spectrograms from a directory of audio files — MI355X path of the reference's
`scripts/audio_to_images.py:17-80` (same CLI: `:83-115`, same on-disk format: a HF `datasets` DatasetDict{"train"} with
features `image: Image()`, `audio_file: string`, `slice: int16`, `:67-78`).

The reference converts one slice at a time on the CPU (`mel.audio_slice_to_image(slice)` in a Python loop, `:44-46`);
here ALL slices of a file go through the batched HIP Mel kernels in one launch (`Mel.audio_slices_to_images`, chunks of
`--batch_slices`), then the reference's silent-slice filter (`:48-51`: every pixel == 255) and PNG encoding run on the host.
Decoding: WAV only (librosa's mp3/m4a decoders and resampler are off the hot path — such files are reported and
skipped, exactly as the reference skips files it cannot load, `:36-42`).
"""
import argparse
import io
import logging
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from audiodiffusion import Mel  # noqa: E402

logging.basicConfig(level=logging.WARN)
logger = logging.getLogger("audio_to_images")


def file_examples(mel, audio_file, batch_slices):
    """All non-silent slices of one file -> list of {"image": {"bytes": png}, "audio_file", "slice"}."""
    from PIL import Image
    mel.load_audio(audio_file)
    n = mel.get_number_of_slices()
    out = []
    for lo in range(0, n, batch_slices):
        idx = list(range(lo, min(n, lo + batch_slices)))
        images = mel.audio_slices_to_images([mel.get_audio_slice(i) for i in idx])     # (B, y_res, x_res) uint8
        for i, img in zip(idx, images):
            assert img.shape == (mel.n_mels, mel.x_res), "Wrong resolution"
            if (img == 255).all():                       # skip completely silent slices (`:48-51`)
                logger.warning("File %s slice %d is completely silent", audio_file, i)
                continue
            with io.BytesIO() as output:
                Image.fromarray(img).save(output, format="PNG")
                out.append({"image": {"bytes": output.getvalue()}, "audio_file": audio_file, "slice": i})
    return out


def main(args):
    mel = Mel(x_res=args.resolution[0], y_res=args.resolution[1], hop_length=args.hop_length,
              sample_rate=args.sample_rate, n_fft=args.n_fft)
    os.makedirs(args.output_dir, exist_ok=True)
    audio_files = sorted(os.path.join(root, file) for root, _, files in os.walk(args.input_dir) for file in files
                         if re.search(r"\.(mp3|wav|m4a)$", file, re.IGNORECASE))
    examples = []
    for audio_file in audio_files:
        try:
            examples.extend(file_examples(mel, audio_file, args.batch_slices))
        except KeyboardInterrupt:
            raise
        except Exception as e:                           # unreadable / unsupported file: report and continue (`:36-42`)
            print(f"{audio_file}: {e}")
            continue
    if len(examples) == 0:
        logger.warning("No valid audio files were found.")
        return None
    import pandas as pd
    from datasets import Dataset, DatasetDict, Features, Image, Value
    ds = Dataset.from_pandas(pd.DataFrame(examples),
                             features=Features({"image": Image(), "audio_file": Value(dtype="string"),
                                                "slice": Value(dtype="int16")}))
    dsd = DatasetDict({"train": ds})
    dsd.save_to_disk(os.path.join(args.output_dir))
    if args.push_to_hub:
        raise NotImplementedError("--push_to_hub needs network access; copy the saved dataset instead")
    return dsd


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description="Create dataset of Mel spectrograms from directory of audio files.")
    parser.add_argument("--input_dir", type=str)
    parser.add_argument("--output_dir", type=str, default="data")
    parser.add_argument("--resolution", type=str, default="256", help="Either square resolution or width,height.")
    parser.add_argument("--hop_length", type=int, default=512)
    parser.add_argument("--push_to_hub", type=str, default=None)
    parser.add_argument("--sample_rate", type=int, default=22050)
    parser.add_argument("--n_fft", type=int, default=2048)
    parser.add_argument("--batch_slices", type=int, default=256, help="slices per Mel kernel launch (not in the reference)")
    args = parser.parse_args(argv)
    if args.input_dir is None:
        raise ValueError("You must specify an input directory for the audio files.")
    try:
        args.resolution = (int(args.resolution), int(args.resolution))
    except ValueError:
        try:
            args.resolution = tuple(int(x) for x in args.resolution.split(","))
            if len(args.resolution) != 2:
                raise ValueError
        except ValueError:
            raise ValueError("Resolution must be a tuple of two integers or a single integer.")
    assert isinstance(args.resolution, tuple)
    return args


if __name__ == "__main__":
    main(parse_args())
