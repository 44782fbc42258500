"""Stand-in for torchvision (see ../README.md): the three transforms the reference's training script composes."""
