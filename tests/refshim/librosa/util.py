import numpy as np


def normalize(x):
    m = np.abs(x).max()
    return x / m if m > 0 else x
