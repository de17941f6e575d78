class engine(object):  # numbers_.py:3,7 — English path unused
    def number_to_words(self, *a, **k):
        raise NotImplementedError
