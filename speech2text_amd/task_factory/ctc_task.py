"""CTC task (reference task_factory/ctc_task.py:32-227): cmvn -> encoder -> decoder -> CTC."""
from speech2text_amd.model.decoder.decoder import Decoder
from speech2text_amd.model.encoder.encoder import Encoder
from speech2text_amd.model.loss.loss import Loss
from speech2text_amd.task_factory.base import TaskBase


class CtcTask(TaskBase):
    def __init__(self, config) -> None:
        super().__init__(config)
        self._encoder = Encoder(config["encoder"])
        self._decoder = Decoder(config["decoder"])
        self._loss = Loss(config["loss"])

    def training_step(self, batch, batch_idx):
        feat, feat_len = self.features(batch)
        enc, enc_len = self._encoder(feat, feat_len)
        dec, dec_len = self._decoder(enc, enc_len)
        loss = self._loss({"logits": dec, "logits_length": dec_len, "targets": batch["label"],
                           "targets_length": batch["label_length"]})
        self.log("train_loss", loss, sync_dist=True, prog_bar=True, logger=True)
        return loss.mean()
