"""Running metrics over an online run -- same names and return shapes as the reference's utils/metric_manager.py."""
import numpy as np


def regression_metric(pred, real):
    """[inf, running mean squared error after 1, 2, ... samples] as an [N+1, 1] array   (reference :7-15)."""
    pred = np.asarray(pred, dtype=np.float64).reshape(-1)
    real = np.asarray(real, dtype=np.float64).reshape(-1)
    n = min(len(pred), len(real))
    running = np.cumsum((pred[:n] - real[:n]) ** 2) / np.arange(1, n + 1)
    return np.concatenate([[np.inf], running]).reshape([-1, 1])


def classfication_metric(pred, real):
    """(logloss_i / (i+1), running accuracy), both [N, 1]   (reference :18-29; the first term is the reference's
    per-sample log(1 + exp(-pred * real)) scaled by 1/(i+1), not a running mean)."""
    pred = np.asarray(pred, dtype=np.float64).reshape(-1)
    real = np.asarray(real, dtype=np.float64).reshape(-1)
    n = min(len(pred), len(real))
    steps = np.arange(1, n + 1)
    metric = np.log(1.0 + np.exp(-pred[:n] * real[:n])) / steps
    metric_acc = np.cumsum(pred[:n] == real[:n]) / steps
    return metric.reshape([-1, 1]), metric_acc.reshape([-1, 1])
