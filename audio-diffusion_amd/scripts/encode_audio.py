#!/usr/bin/env python
"""Create the pickled {audio_file: encoding} table `train_unet.py --encodings` reads — MI355X path of the reference's
`scripts/encode_audio.py:10-41` (same CLI and output format: `encodings[audio_file] = audio_encoder.encode([audio_file])`,
a (1, 100) tensor per track, `pickle.dump` to `--output_file`).

Differences: the AudioEncoder checkpoint is a local directory (`--audio_encoder`; the reference pulls
"teticio/audio-encoder" from the hub, which is unreachable here), and every track's slices run through the batched HIP Mel
kernels and the native encoder in one batch (`AudioEncoder.encode`).
"""
import argparse
import os
import pickle
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from audiodiffusion import AudioEncoder  # noqa: E402


def main(args, audio_encoder=None):
    from datasets import load_dataset, load_from_disk
    if audio_encoder is None:
        audio_encoder = AudioEncoder.from_pretrained(args.audio_encoder)
    if os.path.exists(args.dataset_name):
        dataset = load_from_disk(args.dataset_name)["train"]
    else:
        dataset = load_dataset(args.dataset_name, args.dataset_config_name, cache_dir=args.cache_dir, split="train")
    encodings = {}
    for audio_file in dict.fromkeys(dataset["audio_file"]):          # unique, in first-seen order
        encodings[audio_file] = audio_encoder.encode([audio_file]).cpu()
    os.makedirs(os.path.dirname(os.path.abspath(args.output_file)), exist_ok=True)
    with open(args.output_file, "wb") as f:
        pickle.dump(encodings, f)
    return encodings


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description="Create pickled audio encodings for dataset of audio files.")
    parser.add_argument("--dataset_name", type=str, required=True)
    parser.add_argument("--dataset_config_name", type=str, default=None)
    parser.add_argument("--cache_dir", type=str, default=None)
    parser.add_argument("--output_file", type=str, default="data/encodings.p")
    parser.add_argument("--use_auth_token", type=bool, default=False)
    parser.add_argument("--audio_encoder", type=str, default="teticio/audio-encoder",
                        help="local directory of the AudioEncoder checkpoint (config.json + diffusion_pytorch_model.*)")
    return parser.parse_args(argv)


if __name__ == "__main__":
    main(parse_args())
