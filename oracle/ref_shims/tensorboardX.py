class SummaryWriter(object):  # logger.py:3 — observability stub
    def __init__(self, *a, **k):
        self.scalars = []

    def add_scalar(self, tag, val, step):
        self.scalars.append((tag, float(val), int(step)))

    def add_histogram(self, *a, **k):
        pass

    def add_image(self, *a, **k):
        pass
