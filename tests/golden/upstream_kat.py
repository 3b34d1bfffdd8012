"""Known answers HELD UPSTREAM for the third-party ops the reference delegates to (the reference itself ships no
vectors, and TensorFlow / Keras cannot be imported here).  Every number below is a constant published in the named upstream
test or documentation, typed in from it; none was produced by this repository's code.  tests/test_upstream_kat.py holds the
oracle AND the HIP kernels to them.

What they pin:
  * tf.nn.ctc_loss (the op behind K.ctc_batch_cost, /root/reference/model.py:62-64): loss and gradient w.r.t. the logits;
  * K.ctc_batch_cost itself -- Keras' own backend test runs the same two sequences through the very function the reference
    calls (log(y_pred + 1e-7) -> tf.nn.ctc_loss, blank = last class) and holds the same two losses to atol 1e-5;
  * keras.losses.categorical_crossentropy on probabilities (/root/reference/model.py:350).
Not pinned (no upstream-held constant known to us): keras_layer_normalization with epsilon = K.epsilon()**2 (model.py:4,32-33)
-- its test-suite compares against its own numpy restatement; CuDNNGRU; the fused BatchNormalization.
"""
import numpy as np

# ---- tensorflow/python/kernel_tests/ctc_loss_op_test.py, CTCLossTest.testBasic (TF 1.x): depth 6 (blank = 5), 5 frames.
# "input_prob_matrix_0/1" are fed as logits = log(prob) (the op's internal softmax gives the probabilities back);
# "gradient_log_prob_0/1" are d loss / d logits = softmax - state occupancy; "loss_log_prob_0/1" are -loss.
CTC_TARGETS = [[0, 1, 2, 1, 0], [0, 1, 1, 0]]
CTC_LOSS = [3.34211, 5.42262]                       # -loss_log_prob_0, -loss_log_prob_1 (also keras backend_test.py test_ctc `ref`)
CTC_PROBS = [np.asarray(
    [[0.633766, 0.221185, 0.0917319, 0.0129757, 0.0142857, 0.0260553],
     [0.111121, 0.588392, 0.278779, 0.0055756, 0.00569609, 0.010436],
     [0.0357786, 0.633813, 0.321418, 0.00249248, 0.00272882, 0.0037688],
     [0.0663296, 0.643849, 0.280111, 0.00283995, 0.0035545, 0.00331533],
     [0.458235, 0.396634, 0.123377, 0.00648837, 0.00903441, 0.00623107]], dtype=np.float64), np.asarray(
    [[0.30176, 0.28562, 0.0831517, 0.0862751, 0.0816851, 0.161508],
     [0.24082, 0.397533, 0.0557226, 0.0546814, 0.0557528, 0.19549],
     [0.230246, 0.450868, 0.0389607, 0.038309, 0.0391602, 0.202456],
     [0.280884, 0.429522, 0.0326593, 0.0339046, 0.0326856, 0.190345],
     [0.423286, 0.315517, 0.0338439, 0.0393744, 0.0339315, 0.154046]], dtype=np.float64)]
CTC_GRAD_LOGITS = [np.asarray(
    [[-0.366234, 0.221185, 0.0917319, 0.0129757, 0.0142857, 0.0260553],
     [0.111121, -0.411608, 0.278779, 0.0055756, 0.00569609, 0.010436],
     [0.0357786, 0.633813, -0.678582, 0.00249248, 0.00272882, 0.0037688],
     [0.0663296, -0.356151, 0.280111, 0.00283995, 0.0035545, 0.00331533],
     [-0.541765, 0.396634, 0.123377, 0.00648837, 0.00903441, 0.00623107]], dtype=np.float64), np.asarray(
    [[-0.69824, 0.28562, 0.0831517, 0.0862751, 0.0816851, 0.161508],
     [0.24082, -0.602467, 0.0557226, 0.0546814, 0.0557528, 0.19549],
     [0.230246, 0.450868, 0.0389607, 0.038309, 0.0391602, -0.797544],
     [0.280884, -0.570478, 0.0326593, 0.0339046, 0.0326856, 0.190345],
     [-0.576714, 0.315517, 0.0338439, 0.0393744, 0.0339315, 0.154046]], dtype=np.float64)]
# keras/tests/keras/backend/backend_test.py, test_ctc (Keras 2.2.x): the same two probability matrices as a (2, 5, 6) batch,
# labels [[0,1,2,1,0],[0,1,1,0,-1]], label_lens [5,4], input_lens [5,5] through K.ctc_batch_cost; TensorFlow-backend `ref`
# = [3.34211, 5.42262], assert_allclose(..., atol=1e-05).
KERAS_CTC_ATOL = 1e-5

# ---- tf.keras.losses.categorical_crossentropy docstring example (same math as Keras 2.2.x backend.categorical_crossentropy on
# probabilities: renormalise, clip to [1e-7, 1 - 1e-7], -sum(y * log p)): per-sample losses and their mean as printed there.
CCE_Y_TRUE = np.asarray([[0, 1, 0], [0, 0, 1]], dtype=np.float64)
CCE_Y_PRED = np.asarray([[0.05, 0.95, 0], [0.1, 0.8, 0.1]], dtype=np.float64)
CCE_LOSS = [0.0513, 2.303]                          # as printed there (numpy's 4-digit array repr)
CCE_MEAN = 1.177
CCE_ATOL = 5e-4                                     # half a unit of the last printed digit
