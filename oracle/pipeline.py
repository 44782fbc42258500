"""Oracle restatement of `audiodiffusion/pipeline_audio_diffusion.py:39-258` (TEST INFRASTRUCTURE ONLY).

Follows `AudioDiffusionPipeline.__call__` (:71-205), `encode` (:207-242) and `slerp`
(:244-258) line by line, including the aliasing `images = noise` (:131), the
`images[0, 0] = ...` write (:150) and the numpy half-to-even `round()` (:194).
Runs on torch-CPU with the oracle UNet / schedulers / Mel. Extension for parity
tests only: `step_noise` (list of per-step noise tensors) replaces `randn_tensor`
draws inside scheduler.step so the HIP path can be fed identical noise, and
`audio=False` skips the serial image_to_audio map (:201).
"""
from math import acos, sin

import numpy as np
import torch
from PIL import Image

from .schedulers import DDIMScheduler


class AudioDiffusionPipeline:
    def __init__(self, vqvae, unet, mel, scheduler):
        self.vqvae, self.unet, self.mel, self.scheduler = vqvae, unet, mel, scheduler
        self.device = torch.device("cpu")

    def progress_bar(self, it):
        return it

    def get_default_steps(self):  # :63-69
        return 50 if isinstance(self.scheduler, DDIMScheduler) else 1000

    @torch.no_grad()
    def __call__(self, batch_size=1, audio_file=None, raw_audio=None, slice=0, start_step=0, steps=None,
                 generator=None, mask_start_secs=0, mask_end_secs=0, step_generator=None, eta=0, noise=None,
                 encoding=None, return_dict=True, step_noise=None, audio=True, return_float=False):
        steps = steps or self.get_default_steps()
        self.scheduler.set_timesteps(steps)
        step_generator = step_generator or generator
        if type(self.unet.sample_size) == int:
            self.unet.sample_size = (self.unet.sample_size, self.unet.sample_size)
        if noise is None:
            noise = torch.randn(
                (batch_size, self.unet.in_channels, self.unet.sample_size[0], self.unet.sample_size[1]),
                generator=generator,
            )
        images = noise
        mask = None

        if audio_file is not None or raw_audio is not None:
            self.mel.load_audio(audio_file, raw_audio)
            input_image = self.mel.audio_slice_to_image(slice)
            input_image = np.frombuffer(input_image.tobytes(), dtype="uint8").reshape(
                (input_image.height, input_image.width)
            )
            input_image = (input_image / 255) * 2 - 1
            input_images = torch.tensor(input_image[np.newaxis, :, :], dtype=torch.float)
            if self.vqvae is not None:
                input_images = self.vqvae.encode(torch.unsqueeze(input_images, 0)).latent_dist.sample(
                    generator=generator
                )[0]
                input_images = 0.18215 * input_images
            if start_step > 0:
                images[0, 0] = self.scheduler.add_noise(input_images, noise, self.scheduler.timesteps[start_step - 1])
            pixels_per_second = (
                self.unet.sample_size[1] * self.mel.get_sample_rate() / self.mel.x_res / self.mel.hop_length
            )
            mask_start = int(mask_start_secs * pixels_per_second)
            mask_end = int(mask_end_secs * pixels_per_second)
            mask = self.scheduler.add_noise(input_images, noise, self.scheduler.timesteps[start_step:].clone())

        for step, t in enumerate(self.progress_bar(self.scheduler.timesteps[start_step:])):
            if hasattr(self.unet.config, "get") and self.unet.config.get("cross_attention_dim"):   # :160-161
                model_output = self.unet(images, t, encoding)["sample"]
            else:
                model_output = self.unet(images, t)["sample"]
            vn = None if step_noise is None else step_noise[step]
            if isinstance(self.scheduler, DDIMScheduler):
                images = self.scheduler.step(
                    model_output=model_output, timestep=t, sample=images, eta=eta, generator=step_generator,
                    variance_noise=vn,
                )["prev_sample"]
            else:
                images = self.scheduler.step(
                    model_output=model_output, timestep=t, sample=images, generator=step_generator, variance_noise=vn
                )["prev_sample"]
            if mask is not None:
                if mask_start > 0:
                    images[:, :, :, :mask_start] = mask[:, step, :, :mask_start]
                if mask_end > 0:
                    images[:, :, :, -mask_end:] = mask[:, step, :, -mask_end:]

        if self.vqvae is not None:
            images = 1 / 0.18215 * images
            images = self.vqvae.decode(images)["sample"]

        final_float = images
        images = (images / 2 + 0.5).clamp(0, 1)
        images = images.cpu().permute(0, 2, 3, 1).numpy()
        images = (images * 255).round().astype("uint8")
        images = list(
            map(lambda _: Image.fromarray(_[:, :, 0]), images)
            if images.shape[3] == 1
            else map(lambda _: Image.fromarray(_, mode="RGB").convert("L"), images)
        )
        audios = list(map(lambda _: self.mel.image_to_audio(_), images)) if audio else []
        if return_float:
            return images, final_float
        if not return_dict:
            return images, (self.mel.get_sample_rate(), audios)
        return dict(audios=np.array(audios)[:, np.newaxis, :], images=images)

    @torch.no_grad()
    def encode(self, images, steps=50):  # :207-242
        assert isinstance(self.scheduler, DDIMScheduler)
        self.scheduler.set_timesteps(steps)
        sample = np.array(
            [np.frombuffer(image.tobytes(), dtype="uint8").reshape((1, image.height, image.width)) for image in images]
        )
        sample = (sample / 255) * 2 - 1
        sample = torch.Tensor(sample)
        for t in self.progress_bar(torch.flip(self.scheduler.timesteps, (0,))):
            prev_timestep = t - self.scheduler.config.num_train_timesteps // self.scheduler.num_inference_steps
            alpha_prod_t = self.scheduler.alphas_cumprod[t]
            alpha_prod_t_prev = (
                self.scheduler.alphas_cumprod[prev_timestep]
                if prev_timestep >= 0
                else self.scheduler.final_alpha_cumprod
            )
            beta_prod_t = 1 - alpha_prod_t
            model_output = self.unet(sample, t)["sample"]
            pred_sample_direction = (1 - alpha_prod_t_prev) ** (0.5) * model_output
            sample = (sample - pred_sample_direction) * alpha_prod_t_prev ** (-0.5)
            sample = sample * alpha_prod_t ** (0.5) + beta_prod_t ** (0.5) * model_output
        return sample

    @staticmethod
    def slerp(x0, x1, alpha):  # :244-258
        theta = acos(torch.dot(torch.flatten(x0), torch.flatten(x1)) / torch.norm(x0) / torch.norm(x1))
        return sin((1 - alpha) * theta) * x0 / sin(theta) + sin(alpha * theta) * x1 / sin(theta)
