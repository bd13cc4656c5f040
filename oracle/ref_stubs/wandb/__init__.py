"""Stand-in (wandb is absent offline): logging calls are no-ops."""


def save(*a, **k):
    return None


def log(*a, **k):
    return None


def init(*a, **k):
    return None
