"""fairseq-style registries with the reference's registered names (fairseq itself is not a dependency on the GPU box):
  models     s2t_conformer_dag (s2t_conformer_dag.py:60), s2s_conformer_dag_fastspeech2 (s2s_conformer_dag_fastspeech2.py:42)
  criteria   nat_dag_loss (nat_dag_loss.py:45), s2s_dag_fastspeech2_loss (s2s_dag_fastspeech2_loss.py:26)
  tasks      nat_speech_to_text (nat_speech_to_text.py:30), nat_speech_to_speech (nat_speech_to_speech.py:32)
  generators S2SNATGenerator (generator/s2s_nat_generator.py:23)"""
from . import criterions, synthetic
from .generator import S2SNATGenerator
from .models.daspeech import S2SConformerDAGFastSpeech2Model, S2TConformerDAGModel

MODEL_REGISTRY = {"s2t_conformer_dag": S2TConformerDAGModel, "s2s_conformer_dag_fastspeech2": S2SConformerDAGFastSpeech2Model}
CRITERION_REGISTRY = {"nat_dag_loss": criterions.NATDAGLoss, "s2s_dag_fastspeech2_loss": criterions.S2SDAGFastSpeech2Loss}
TASK_REGISTRY = {"nat_speech_to_text": synthetic.NATSpeechToTextTask, "nat_speech_to_speech": synthetic.NATSpeechToSpeechTask}
GENERATOR_REGISTRY = {"nat_s2s": S2SNATGenerator}
