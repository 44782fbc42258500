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


AUDIO_SUFFIX = re.compile(r"\.(mp3|wav|m4a)$", re.IGNORECASE)      # the extensions the reference walks for (`:24-30`)


def find_audio(input_dir):
    return sorted(os.path.join(d, f) for d, _, names in os.walk(input_dir) for f in names if AUDIO_SUFFIX.search(f))


def write_dataset(examples, output_dir):
    """Rows -> HF DatasetDict{"train"} on disk with the reference's schema (`:67-78`)."""
    import pandas as pd
    from datasets import Dataset, DatasetDict, Features, Image, Value
    schema = Features({"image": Image(), "audio_file": Value(dtype="string"), "slice": Value(dtype="int16")})
    dsd = DatasetDict({"train": Dataset.from_pandas(pd.DataFrame(examples), features=schema)})
    dsd.save_to_disk(os.path.join(output_dir))
    return dsd


def main(args):
    width, height = args.resolution
    mel = Mel(x_res=width, y_res=height, hop_length=args.hop_length, sample_rate=args.sample_rate, n_fft=args.n_fft)
    os.makedirs(args.output_dir, exist_ok=True)
    examples = []
    for audio_file in find_audio(args.input_dir):
        try:
            examples += file_examples(mel, audio_file, args.batch_slices)
        except KeyboardInterrupt:
            raise
        except Exception as e:                           # unreadable / unsupported file: report and go on (`:36-42`)
            print(f"{audio_file}: {e}")
    if not examples:
        logger.warning("No valid audio files were found.")
        return None
    dsd = write_dataset(examples, args.output_dir)
    if args.push_to_hub:
        raise NotImplementedError("--push_to_hub needs network access; copy the saved dataset instead")
    return dsd


def _resolution(text):
    """"256" -> (256, 256); "216,96" -> (216, 96) = (width, height), as the reference's --resolution (`:97-110`)."""
    parts = [p for p in str(text).split(",")]
    try:
        nums = [int(p) for p in parts]
    except ValueError:
        nums = []
    if len(nums) == 1:
        return (nums[0], nums[0])
    if len(nums) == 2:
        return (nums[0], nums[1])
    raise ValueError("Resolution must be a tuple of two integers or a single integer.")


FLAGS = (   # the reference's CLI (`:83-95`) plus --batch_slices
    ("--input_dir", dict(type=str)),
    ("--output_dir", dict(type=str, default="data")),
    ("--resolution", dict(type=str, default="256", help="Either square resolution or width,height.")),
    ("--hop_length", dict(type=int, default=512)),
    ("--push_to_hub", dict(type=str, default=None)),
    ("--sample_rate", dict(type=int, default=22050)),
    ("--n_fft", dict(type=int, default=2048)),
    ("--batch_slices", dict(type=int, default=256, help="slices per Mel kernel launch (not in the reference)")),
)


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description="Create dataset of Mel spectrograms from directory of audio files.")
    for flag, kw in FLAGS:
        parser.add_argument(flag, **kw)
    args = parser.parse_args(argv)
    if args.input_dir is None:
        raise ValueError("You must specify an input directory for the audio files.")
    args.resolution = _resolution(args.resolution)
    return args


if __name__ == "__main__":
    main(parse_args())
