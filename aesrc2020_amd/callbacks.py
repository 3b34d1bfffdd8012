"""Minimal Keras-callback counterparts for the ones train.py:80-84 wires into fit_generator
(ReduceLROnPlateau, EarlyStopping, CSVLogger, Callback).  Caller-side plumbing, not on the hot path."""
import csv
import numpy as np


class Callback:
    def set_model(self, model):
        self.model = model

    def on_train_begin(self, logs=None): pass
    def on_train_end(self, logs=None): pass
    def on_epoch_begin(self, epoch, logs=None): pass
    def on_epoch_end(self, epoch, logs=None): pass


class CallbackList:
    def __init__(self, cbs, model):
        self.cbs = list(cbs)
        for c in self.cbs:
            c.set_model(model)

    def __getattr__(self, name):
        def call(*a, **k):
            for c in self.cbs:
                getattr(c, name)(*a, **k)
        return call


class _Monitor(Callback):
    def __init__(self, monitor, mode, min_delta):
        self.monitor, self.min_delta = monitor, abs(min_delta)
        self.sign = 1.0 if (mode == "max" or (mode == "auto" and "acc" in monitor)) else -1.0
        self.best = -np.inf; self.wait = 0

    def improved(self, logs):
        cur = (logs or {}).get(self.monitor)
        if cur is None:
            return None
        if self.sign * cur > self.best + self.min_delta:
            self.best = self.sign * cur; self.wait = 0
            return True
        self.wait += 1
        return False


class ReduceLROnPlateau(_Monitor):
    def __init__(self, monitor="val_loss", factor=0.1, patience=10, verbose=0, mode="auto", min_delta=1e-4, cooldown=0, min_lr=0):
        super().__init__(monitor, mode, min_delta)
        self.factor, self.patience, self.verbose, self.cooldown, self.min_lr = factor, patience, verbose, cooldown, min_lr
        self.cool = 0

    def on_epoch_end(self, epoch, logs=None):
        if self.cool > 0:
            self.cool -= 1; self.wait = 0
        if self.improved(logs) is False and self.wait >= self.patience:
            old = self.model.store.get_lr()
            if old > self.min_lr:
                new = max(old * self.factor, self.min_lr)
                self.model.store.set_lr(new)
                if self.verbose:
                    print("\nEpoch %05d: ReduceLROnPlateau reducing learning rate to %s." % (epoch + 1, new))
                self.cool = self.cooldown; self.wait = 0


class EarlyStopping(_Monitor):
    def __init__(self, monitor="val_loss", min_delta=0, patience=0, verbose=0, mode="auto"):
        super().__init__(monitor, mode, min_delta)
        self.patience, self.verbose = patience, verbose

    def on_epoch_end(self, epoch, logs=None):
        if self.improved(logs) is False and self.wait >= self.patience:
            self.model.stop_training = True
            if self.verbose:
                print("Epoch %05d: early stopping" % (epoch + 1))


class CSVLogger(Callback):
    def __init__(self, filename):
        self.filename = filename; self.keys = None

    def on_epoch_end(self, epoch, logs=None):
        logs = logs or {}
        if self.keys is None:
            self.keys = sorted(logs)
            with open(self.filename, "w", newline="") as f:
                csv.writer(f).writerow(["epoch"] + self.keys)
        with open(self.filename, "a", newline="") as f:
            csv.writer(f).writerow([epoch] + [logs.get(k, "") for k in self.keys])
