/*
 *  FeatureExtractionImpl.scala -- replacement of the reference's class of this name: the scsynth non-realtime
 *  render is replaced by the library's STFT -> mel -> DCT + loudness kernels (same framing and file formats; the
 *  spectral model is the library's own, so databases extracted by the two are not interchangeable).
 */
package de.sciss.strugatzki
package impl

import de.sciss.processor.impl.ProcessorImpl

import scala.concurrent.blocking

private[strugatzki] final class FeatureExtractionImpl(val config: FeatureExtraction.Config)
  extends FeatureExtraction with ProcessorImpl[Unit, FeatureExtraction] {

  protected def body(): Unit = blocking {
    Native.withContexts(Native.devices.take(1)) { ctx =>
      Native.run(this, p => { progress = p }, () => checkAborted())(Native.runExtraction(ctx(0), config.toXML.toString, _))
    }
  }
}
