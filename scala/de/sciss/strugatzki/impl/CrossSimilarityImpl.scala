/*
 *  CrossSimilarityImpl.scala -- replacement of the reference's class of this name (the shorter span slid along the
 *  longer one by the exact correlation kernel; output type AIFF).
 */
package de.sciss.strugatzki
package impl

import de.sciss.processor.impl.ProcessorImpl

import scala.concurrent.blocking

private[strugatzki] final class CrossSimilarityImpl(val config: CrossSimilarity.Config)
  extends CrossSimilarity with ProcessorImpl[CrossSimilarity.Product, CrossSimilarity] {

  protected def body(): Unit = blocking {
    Native.withContexts(Native.devices.take(1)) { ctx =>
      Native.run(this, p => { progress = p }, () => checkAborted())(Native.runCrossSimilarity(ctx(0), config.toXML.toString, _))
    }
  }
}
